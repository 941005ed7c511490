#!/usr/bin/env python3
"""Secondary benchmark: the full three-stage pipeline (BASELINE.json configs[2] shape:
~3.6 k documents, stage 1 top-1000 -> stage 2 keep 100 -> stage 3 top-10, bf16) on
one MI355X with RANDOMLY INITIALISED models of the reference's architectures (no
weights exist offline), i.e. throughput and stage shares only — not a quality run.
bench.py (stage-1 at 10M x 768) stays the headline measurement.

    python bench_pipeline.py [--docs 3633] [--queries 16] [--cache]
    python bench_pipeline.py --gpus N --store --ids --many 64 [--docs D]    # row-sharded over N ranks (starts them itself)

With --gpus N > 1 the pipeline is parallel_pipeline.ShardedRetrievalPipeline: corpus rows, stage-2 token store,
stage-3 token ids, BM25 postings and text row-sharded over N ranks (RCCL; TS_BENCH_BACKEND=gloo rehearses the same
code path with ranks sharing GPUs), every rank issues the same search_many calls, the time is the maximum over ranks.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=3633)
    ap.add_argument("--queries", type=int, default=16)
    ap.add_argument("--stage1", default="random:bert")          # 768-d bi-encoder stand-in
    ap.add_argument("--stage2", default="random:modernbert")    # GTE-ModernColBERT backbone shape
    ap.add_argument("--stage3", default="random:minilm")        # ms-marco-MiniLM-L6 shape
    ap.add_argument("--cache", action="store_true", help="cache stage-2 token matrices per document")
    ap.add_argument("--store", action="store_true",
                    help="stage-2 token store filled at add time, read in place by ts_maxsim_indexed")
    ap.add_argument("--graphs", action="store_true", help="replay the query forwards (one query, or a search_many batch padded to a bucket) from HIP graphs")
    ap.add_argument("--bm25", action="store_true", help="BM25 + RRF fusion in stage 1, like the reference's default")
    ap.add_argument("--cprofile", action="store_true", help="print the host-side hot spots of the timed region (stderr)")
    ap.add_argument("--ids", action="store_true",
                    help="stage-3 token ids cached at add time: search_many runs every stage on arrays (needs --store --many)")
    ap.add_argument("--s3-batch", type=int, default=1024, help="pairs per cross-encoder forward in search_many")
    ap.add_argument("--no-lean", action="store_true", help="stage 3 through the transformers module instead of the written-out forward")
    ap.add_argument("--torch-attention", action="store_true", help="stage 3 attention through torch's masked SDPA instead of ts_attention_varlen")
    ap.add_argument("--tune-gemms", action="store_true", help="PyTorch TunableOp for the GEMMs (stage-3 widths padded to multiples of 16; "
                    "one untimed pass over the queries first, so that every shape is tuned before the timed region)")
    ap.add_argument("--keep", action="store_true", help="save_intermediate_results (all three record lists are built)")
    ap.add_argument("--gpus", type=int, default=1, help="ranks of a row-sharded pipeline (one process per GPU)")
    ap.add_argument("--many", type=int, default=0,
                    help="queries per RetrievalPipeline.search_many call (every stage batched); 0 = search() per query")
    return ap.parse_args(argv)


def run(args):
    """Build the pipeline, index the synthetic corpus, time the queries; returns the result record.
    bench.py calls this for its `pipeline_search_many_qps` legs (same settings as the command line)."""
    import numpy as np
    import torch
    from tristage_rag_amd.retrieval_pipeline import PipelineConfig, RetrievalPipeline
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    backend = os.environ.get("TS_BENCH_BACKEND", "nccl")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if backend != "nccl":
            local_rank %= max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    rng = np.random.default_rng(0)
    vocab = [f"w{i}" for i in range(5000)]
    docs = [" ".join(rng.choice(vocab, size=int(rng.integers(40, 160)))) for _ in range(args.docs)]
    queries = [" ".join(rng.choice(vocab, size=int(rng.integers(4, 16)))) for _ in range(args.queries)]
    pc = PipelineConfig(stage1_model=args.stage1, stage2_model=args.stage2, stage3_model=args.stage3,
                        device="cuda", cache_dir="/tmp/ts_models", index_dir="/tmp/ts_index",
                        log_file="/tmp/ts_pipeline.log", log_level="WARNING",
                        stage1_top_k=1000, stage2_top_k=100, stage3_top_k=10, stage1_enable_bm25=args.bm25,
                        stage1_index_dtype="f16", stage1_batch_size=64, stage2_batch_size=64,
                        stage3_batch_size=64, stage2_cache_document_embeddings=args.cache,
                        stage2_precompute_document_embeddings=args.store, use_hip_graphs=args.graphs,
                        stage3_cache_document_tokens=args.ids, save_intermediate_results=args.keep,
                        stage3_many_batch_size=args.s3_batch, tune_gemms=args.tune_gemms,
                        stage3_width_multiple=16 if args.tune_gemms else 1)
    if world > 1:
        from tristage_rag_amd.parallel_pipeline import ShardedRetrievalPipeline
        pc.log_file = f"/tmp/ts_pipeline_rank{rank}.log"
        p = ShardedRetrievalPipeline(config=pc)
    else:
        p = RetrievalPipeline(config=pc)
    p.initialize_stages()
    if args.no_lean and hasattr(p.stage3.model, "lean_forward"):
        p.stage3.model.lean_forward = False
    if args.torch_attention and hasattr(p.stage3.model, "_lean_model") and p.stage3.model._lean_model():
        p.stage3.model._lean_model().fused_attention = False
    t0 = time.perf_counter()
    p.add_documents(docs)
    torch.cuda.synchronize()
    t_index = time.perf_counter() - t0
    p.search(queries[0])                      # warm-up
    if args.cache:
        for q in queries:                     # fill the cache (a real deployment fills it at add time)
            p.search(q)
    torch.cuda.synchronize()
    if args.many:
        p.search_many(queries[: args.many])   # warm-up of the batched shapes
        torch.cuda.synchronize()
    t_tune = None
    if args.tune_gemms:                       # every GEMM shape of the run meets the tuner once, untimed
        tt = time.perf_counter()
        if args.many:
            for s in range(0, len(queries), args.many):
                p.search_many(queries[s: s + args.many])
        else:
            for q in queries:
                p.search(q)
        torch.cuda.synchronize()
        t_tune = time.perf_counter() - tt
    prof = None
    if args.cprofile:
        import cProfile
        prof = cProfile.Profile()
        prof.enable()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    if args.many:
        outs = []
        for s in range(0, len(queries), args.many):
            outs.extend(p.search_many(queries[s: s + args.many]))
    else:
        outs = [p.search(q) for q in queries]
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if prof is not None:
        import pstats
        prof.disable()
        pstats.Stats(prof, stream=sys.stderr).sort_stats("cumulative").print_stats(28)
    tm = {k: float(np.mean([o["timing"][k] for o in outs])) for k in ("stage1_time", "stage2_time", "stage3_time", "total_time")}
    g1 = getattr(p.stage1.model, "_graphed", None)
    g2 = getattr(p.stage2, "_graphed", None)
    g3 = getattr(p.stage3.model, "_graphed", None)
    graph_state = {"stage1": None if g1 is None else {"buckets": sorted(g1._graphs), "eager_fallback": g1._broken},
                   "stage2": None if g2 is None else {"buckets": sorted(g2._graphs), "eager_fallback": g2._broken},
                   "stage3": None if g3 is None else {"buckets": sorted(g3._graphs), "eager_fallback": g3._broken}}
    shard = p.shard_info() if world > 1 else None
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return {
        "metric": "full 3-stage pipeline queries/sec (random-init models, throughput only)",
        "value": round(len(queries) / dt, 3), "unit": "queries/s", "n_gpus": world,
        "backend": ("rccl" if backend == "nccl" else backend + " (REHEARSAL: ranks may share GPUs)") if world > 1 else "none",
        "rank0_shard": shard,
        "config": {"workload": f"{args.docs} synthetic docs, S1 top-1000 -> S2 keep 100 -> S3 top-10, bf16",
                   "stage1": args.stage1, "stage2": args.stage2, "stage3": args.stage3,
                   "stage2_token_cache": args.cache, "stage2_token_store": args.store, "hip_graphs": args.graphs,
                   "queries_per_search_many": args.many, "bm25_rrf": args.bm25,
                   "stage3_token_id_cache": args.ids, "stage3_pairs_per_forward": args.s3_batch,
                   "stage3_pairs_per_packed_forward": int(getattr(p.stage3.config, "many_packed_batch_size", args.s3_batch)),
                   "stage3_lean_forward": bool(getattr(p.stage3.model, "_lean", None)), "save_intermediate_results": args.keep,
                   "array_path": bool(args.ids and args.store and getattr(p.stage3, "_pairs_usable", False)
                                      and (args.many or p.config.search_on_arrays))},
        "index_build_s": round(t_index, 3), "gemm_tuning_pass_s": (round(t_tune, 1) if t_tune is not None else None),
        "mean_stage_seconds": {k: round(v, 5) for k, v in tm.items()},
        "hip_graph_state": graph_state,
        "data": "synthetic"}


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:      # become the launcher (nothing here has touched HIP yet)
        import bench
        backend = os.environ.get("TS_BENCH_BACKEND", "nccl")
        if backend == "nccl" and bench.visible_gpus() < args.gpus:
            sys.stderr.write(f"bench_pipeline.py: --gpus {args.gpus} but fewer GPUs are visible\n")
            return 2
        return bench.spawn_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:])
    if args.gpus != int(os.environ.get("WORLD_SIZE", "1")):
        sys.stderr.write("bench_pipeline.py: --gpus does not match WORLD_SIZE\n")
        return 2
    out = run(args)
    if int(os.environ.get("RANK", "0")) == 0:
        print(json.dumps(out), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main() or 0)
