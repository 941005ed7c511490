import cProfile, pstats, sys, io, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tristage_rag_amd.retrieval_pipeline import PipelineConfig, RetrievalPipeline
rng = np.random.default_rng(0)
vocab = [f"w{i}" for i in range(5000)]
docs = [" ".join(rng.choice(vocab, size=int(n))) for n in rng.integers(20, 160, size=3633)]
pc = PipelineConfig(stage1_model="random:bert", stage2_model="random:modernbert", stage3_model="random:minilm", device="cuda",
                    stage1_top_k=1000, stage2_top_k=100, stage3_top_k=10, stage1_enable_bm25=False,
                    stage2_precompute_document_embeddings=True, stage3_cache_document_tokens=True, index_dir="/tmp/ts_idx", cache_dir="/tmp/ts_models")
p = RetrievalPipeline(config=pc); p.initialize_stages()
p.add_documents(docs[:64]); torch.cuda.synchronize()
p2 = RetrievalPipeline(config=pc); p2.initialize_stages()
pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
p2.add_documents(docs); torch.cuda.synchronize()
pr.disable(); print("add_documents s:", round(time.perf_counter() - t0, 3))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(30); print(s.getvalue()[:6000])
