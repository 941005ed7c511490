#!/usr/bin/env python3
"""bench.py — stage-1 brute-force top-k throughput on MI355X (the hot path of
BASELINE.json's north_star; workload = configs[3]: synthetic 10M x 768 fp16
corpus, batch-64 queries, top-1000, row-sharded over N GPUs with one RCCL
all-gather + merge per batch).

    python bench.py --gpus N --steps K --warmup W

A "step" = one batch of 64 queries searched against the whole corpus: local
shard scan + exact top-k on every rank, all-gather of the partial lists, merge.
Corpus and queries are resident in HBM before the timed region.  The total
corpus is fixed as N grows (strong scaling: the metric is quoted "@10Mx768
corpus, 1/2/4/8 GPU").

One JSON line on rank 0.  Besides the driver's contract it carries
  roofline      the fused scan+filter kernel: algorithmic bytes per launch
                (shard rows x padded dim x 2 B, the corpus read once) / its mean
                duration measured with HIP events on the launch stream
  cpu_baseline  the same search on the host cores (oracle.ip_topk_blas: blocked
                SGEMM + partial sort, what faiss-cpu IndexFlatIP does) on a
                bounded row sample, scaled to the full corpus.  N=1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy ceiling)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=10_000_000, help="total corpus rows")
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--k", type=int, default=1000)
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sync", action="store_true",
                    help="wait for every batch before issuing the next (default: batches are enqueued "
                         "back to back and completed + verified by finish() inside the timed region)")
    ap.add_argument("--pipeline", choices=["auto", "on", "off"], default="auto",
                    help="overlap the small kernels of neighbouring batches with the scan (internal streams). "
                         "auto: on for N>1 (short per-rank scans, where the fixed part matters: +10 %% at "
                         "1.25 M rows per rank), off for N=1 (no gain at 10 M rows and the overlapped scan "
                         "runs 2-3 %% slower)")
    ap.add_argument("--no-profile", action="store_true", help="no HIP-event timing of the scan kernel")
    ap.add_argument("--force-exchange", action="store_true",
                    help="diagnostic: with one rank, still run the RCCL all-gather + merge every step")
    ap.add_argument("--submit-stream", choices=["auto", "null", "side"], default="auto",
                    help="stream the searches are submitted from.  auto: a non-default stream when pipelining "
                         "(the NULL stream synchronises implicitly with every blocking stream of the process: "
                         "0.332 -> 0.324 ms per batch at 1.25 M rows per rank), the NULL stream otherwise")
    ap.add_argument("--cpu-sample-rows", type=int, default=400_000)
    ap.add_argument("--one-launch", action="store_true",
                    help="A/B: the one-launch scan (query image + thresholds + scan+filter in one kernel) also where the "
                         "five-launch path is the default (pipelined submission, corpora above 4 M rows)")
    ap.add_argument("--classic", action="store_true",
                    help="A/B: the five-launch filter path (query prep, sample scan, thresholds, scan+filter, select) "
                         "instead of the one-launch scan")
    ap.add_argument("--step-events", choices=["auto", "on", "off"], default="auto",
                    help="one HIP event per timed step on the submitting stream -> ms_per_step_min/max "
                         "(a few microseconds each in-stream).  auto: on for N=1, off for N>1")
    ap.add_argument("--no-encode-leg", action="store_true",
                    help="skip the secondary leg (bi-encoder forward of the 64 query texts + the same search)")
    ap.add_argument("--encoder", default=None,
                    help="bi-encoder of the secondary leg (default: random:bert with hidden = --dim)")
    return ap.parse_args()


def gen_rows(torch, n, d, seed, dtype, device):
    """Synthetic unit-norm rows, generated on the GPU in blocks."""
    g = torch.Generator(device=device).manual_seed(seed)
    x = torch.randn((n, d), generator=g, device=device, dtype=torch.float32)
    x = x / (x.norm(dim=1, keepdim=True) + 1e-8)
    return x.to(dtype)


def cpu_baseline(args, torch):
    import numpy as np
    from oracle import oracle
    n = min(args.cpu_sample_rows, args.rows)
    rng = np.random.default_rng(1234)
    c = rng.standard_normal((n, args.dim), dtype=np.float32)
    c /= (np.linalg.norm(c, axis=1, keepdims=True) + 1e-8)
    q = rng.standard_normal((args.batch, args.dim), dtype=np.float32)
    q /= (np.linalg.norm(q, axis=1, keepdims=True) + 1e-8)
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:  # the SGEMM runs in numpy's BLAS: report the threads THAT pool uses (it may cap below the core count)
        from threadpoolctl import threadpool_info
        blas = [int(p["num_threads"]) for p in threadpool_info() if p.get("user_api") == "blas"]
        if blas:
            cores = min(cores, max(blas))
    except Exception:
        pass
    k = min(args.k, n)
    oracle.ip_topk_blas(c, q, k)  # warm-up (BLAS thread pool, page faults)
    reps, t_total = 0, 0.0
    while t_total < 10.0 and reps < 50:
        t0 = time.perf_counter()
        oracle.ip_topk_blas(c, q, k)
        t_total += time.perf_counter() - t0
        reps += 1
    t = t_total / reps
    scale = args.rows / n
    return {
        "value": args.batch / (t * scale),
        "unit": "queries/s",
        "cores": cores,
        "kind": "port",
        "sample": (f"{n} of {args.rows} rows x {args.dim} fp32, batch {args.batch}, top-{k}; "
                   f"{reps} reps of {t * 1e3:.1f} ms, scaled x{scale:.1f} to the full corpus"),
    }


def encode_leg(args, torch, index, device, tdt, steps, world):
    """Secondary, separately reported: the same stage-1 search fed by the bi-encoder instead of
    pre-encoded queries — tokenise 64 synthetic query texts, one bf16 forward of a randomly
    initialised BERT-base-shaped encoder (no weights exist offline), x/(|x|+1e-8), cast to the
    index dtype, search.  Every rank encodes the (replicated) queries itself, as in §8e."""
    import numpy as np
    from tristage_rag_amd.encoders import SentenceEncoder
    if args.dim % 64:
        return None
    spec = args.encoder or f"random:bert:{args.dim}:12:{args.dim // 64}"
    enc = SentenceEncoder(spec, device=str(device))
    rng = np.random.default_rng(99)
    vocab = [f"w{i}" for i in range(5000)]
    texts = [[" ".join(rng.choice(vocab, size=int(rng.integers(4, 16)))) for _ in range(args.batch)] for _ in range(4)]

    def one(i, async_):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            e = enc.encode(texts[i % 4], batch_size=args.batch, convert_to_numpy=False, convert_to_tensor=True)
        q = (e / (e.norm(dim=1, keepdim=True) + 1e-8)).to(tdt)
        return index.search(q, args.k, async_=True) if async_ else index.search(q, args.k)

    for i in range(2):
        one(i, False)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        one(i, True)
    index.finish()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    te = time.perf_counter()
    for i in range(4):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            enc.encode(texts[i % 4], batch_size=args.batch, convert_to_numpy=False, convert_to_tensor=True)
    torch.cuda.synchronize()
    enc_ms = (time.perf_counter() - te) / 4 * 1e3
    return {"encode_plus_stage1_qps": round(args.batch * steps / dt, 2), "steps": steps,
            "ms_per_step": round(dt / steps * 1e3, 4), "encode_ms_per_batch": round(enc_ms, 4),
            "encoder": spec + " (random init, hash tokenizer, bf16 autocast)"}


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # TS_BENCH_BACKEND=gloo: rehearsal of the N > 1 code path on a box with fewer GPUs than ranks
    # (ranks share devices, the exchange is staged through the host); never the measured setup
    backend = os.environ.get("TS_BENCH_BACKEND", "nccl")
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend != "nccl":
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)
        local_rank = 0
        if args.force_exchange:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29577")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    if args.gpus != world:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    device = torch.device("cuda", local_rank)
    tdt = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[args.dtype]

    from tristage_rag_amd.index import FlatIPIndex
    from tristage_rag_amd.sharded import ShardedFlatIPIndex, shard_bounds

    # ---- build the (sharded) index; nothing here is timed
    lo, hi = shard_bounds(args.rows, world, rank)
    local = FlatIPIndex(args.dim, dtype=args.dtype, device=local_rank)
    local.classic_filter = args.classic or bool(os.environ.get("TS_BENCH_CLASSIC"))
    local.one_launch_filter = args.one_launch
    local.reserve(max(hi - lo, 1))
    blk = 500_000
    for r0 in range(lo, hi, blk):
        n = min(blk, hi - r0)
        local.add(gen_rows(torch, n, args.dim, 1234 + r0 // blk, tdt, device))
    if world > 1 or args.force_exchange:
        index = ShardedFlatIPIndex(args.dim, args.rows, dtype=args.dtype, device=local_rank,
                                   local_index=local)
        index.always_exchange = args.force_exchange
    else:
        index = local
    queries = [gen_rows(torch, args.batch, args.dim, 4321 + i, tdt, device) for i in range(4)]
    torch.cuda.synchronize()

    pipeline = (args.pipeline == "on") or (args.pipeline == "auto" and (world > 1 or args.force_exchange))

    def step(i):
        if args.sync:
            return index.search(queries[i % len(queries)], args.k)
        # the query tensors were materialised (and synchronised) before the loop
        return index.search(queries[i % len(queries)], args.k, async_=True, inputs_ready=pipeline)

    def finish():
        if not args.sync:
            index.finish()

    if args.submit_stream == "side" or (args.submit_stream == "auto" and pipeline):
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        torch.cuda.set_stream(side)
    for i in range(args.warmup):
        step(i)
    finish()
    # per-batch latency with a host sync after every batch (reported, not the metric)
    torch.cuda.synchronize()
    tl = time.perf_counter()
    for i in range(5):
        index.search(queries[i % len(queries)], args.k)
    torch.cuda.synchronize()
    sync_latency_ms = (time.perf_counter() - tl) / 5 * 1e3
    # every 4th batch is bracketed by HIP events on the scan stream (a timing event costs ~5 us
    # in-stream; the sampled launches are still inside the timed region)
    local.set_profiling(not args.no_profile, every=1 if args.sync else 4)
    local.timings(reset=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    step_events = args.step_events == "on" or (args.step_events == "auto" and world == 1)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)] if step_events else []
    t0 = time.perf_counter()
    if evs:
        evs[0].record()
    for i in range(args.steps):
        D, I = step(i)
        if evs:
            evs[i + 1].record()   # on the submitting stream, which every batch's result is ordered on
    finish()                     # completes AND verifies every batch of the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    step_ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps)] if evs else []
    tm = local.timings(reset=True)
    info = local.last_search_info()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- roofline of the dominant kernel (this rank's shard)
    esize = 4 if args.dtype == "f32" else 2
    gran = 64 if args.dtype == "f32" else 128
    dpad = -(-args.dim // gran) * gran
    shard_rows = hi - lo
    alg_bytes = float(-(-shard_rows // 32) * 32) * dpad * esize
    scan_ms, scan_cnt = tm["filter_scan"] if tm["filter_scan"][1] else tm["dense"]
    kernel = (("fused_kernel (query image + thresholds + scan+filter)" if info.get("one_launch") else "scan_kernel<filter>")
              if tm["filter_scan"][1] else "dense path (scan+select)")
    roof = None
    if scan_cnt:
        avg_ms = scan_ms / scan_cnt
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(f"{args.rows // world}x{args.dim}x{args.dtype}")
            except Exception:
                traffic = None
        roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "kernel": kernel,
                "avg_kernel_ms": round(avg_ms, 4), "launches": scan_cnt,
                "algorithmic_bytes_per_launch": alg_bytes}

    enc_leg = None
    if not args.no_encode_leg:
        try:
            enc_leg = encode_leg(args, torch, index, device, tdt, max(4, min(args.steps, 20)), world)
        except Exception as e:  # the secondary leg must never take the headline measurement down
            enc_leg = {"error": repr(e)}

    def _rows_label(n):
        return f"{n // 1_000_000}M" if n % 1_000_000 == 0 and n >= 1_000_000 else (
            f"{n / 1e6:g}M" if n >= 1_000_000 else str(n))

    if rank == 0:
        out = {
            "metric": (f"end-to-end queries/sec @{_rows_label(args.rows)}x{args.dim} corpus "
                       "(stage-1 exact top-k, pre-encoded HBM-resident queries)"),
            "value": round(args.batch * args.steps / elapsed, 2),
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "ms_per_step_min": round(min(step_ms), 4) if step_ms else None,
            "ms_per_step_max": round(max(step_ms), 4) if step_ms else None,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": (f"synthetic {args.rows}x{args.dim} {args.dtype} corpus, batch-{args.batch} "
                                    f"queries, stage-1 exact top-{args.k}, row-sharded over {world} GPU(s)"
                                    + ((", RCCL all-gather + HIP merge" if backend == "nccl" else
                                        f", REHEARSAL: {backend} group, ranks share devices, host-staged exchange")
                                       if world > 1 else "")),
                       "rows": args.rows, "dim": args.dim, "batch": args.batch, "k": args.k,
                       "submission": "synchronous per batch" if args.sync else
                                     ("batches enqueued back to back (async" + ("" if not pipeline else ", pipelined: prep/select of "
                                      "neighbouring batches overlap the scan") + "), verified by finish() in the timed region"),
                       "sync_batch_latency_ms": round(sync_latency_ms, 4),
                       "search_path": info["path"] + (" (one launch: query image + thresholds + scan+filter)"
                                                      if info.get("one_launch") else ""),
                       "max_candidates_per_query": info["max_candidates"],
                       "phase_ms_per_step": {p: round(v[0] / max(v[1], 1), 4) for p, v in tm.items() if v[1]}},
            "roofline": roof,
            "secondary": enc_leg,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, torch)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
