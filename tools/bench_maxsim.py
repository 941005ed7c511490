#!/usr/bin/env python3
"""Stage-2 MaxSim over a resident token store: time per query and achieved HBM rate.

    python tools/bench_maxsim.py [--docs 1000] [--lq 32] [--hidden 768] [--dtype bf16] [--len-lo 64 --len-hi 192]

Algorithmic bytes per query = (sum(Ld) + Lq) * H * esize: every candidate's
token rows read once.  Timed with events on the stream the kernels are launched on (torch's
current stream is what index.maxsim_indexed passes to the C ABI)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=1000)
    ap.add_argument("--lq", type=int, default=32)
    ap.add_argument("--hidden", type=int, default=768)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--len-lo", type=int, default=64)
    ap.add_argument("--len-hi", type=int, default=192)
    ap.add_argument("--store-docs", type=int, default=200_000, help="documents in the resident store")
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--mode", default="maxsim")
    ap.add_argument("--batch", type=int, default=0,
                    help="score this many queries per launch (ts_maxsim_indexed_batch) instead of one")
    ap.add_argument("--no-check", action="store_true", help="batch mode: skip the comparison with per-query launches")
    ap.add_argument("--contiguous", action="store_true",
                    help="diagnostic: candidates are neighbours in the store (no scattered 2 MiB pages)")
    args = ap.parse_args()
    import torch
    from tristage_rag_amd.index import maxsim_indexed, maxsim_indexed_batch
    tdt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.dtype]
    esize = 4 if args.dtype == "f32" else 2
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(7)
    lens_all = torch.randint(args.len_lo, args.len_hi + 1, (args.store_docs,), generator=g, device=dev, dtype=torch.int64)
    starts_all = torch.cumsum(lens_all, 0) - lens_all
    rows = int(lens_all.sum().item())
    store = torch.randn((rows, args.hidden), generator=g, device=dev, dtype=torch.float32).to(tdt)
    q = torch.randn((args.lq, args.hidden), generator=g, device=dev, dtype=torch.float32).to(tdt)
    # a different random candidate set per repetition: nothing is cache-resident between queries
    if args.contiguous:
        offs = torch.randint(0, args.store_docs - args.docs, (args.reps + 3,), generator=g, device=dev).tolist()
        picks = [torch.arange(o, o + args.docs, device=dev) for o in offs]
    else:
        picks = [torch.randperm(args.store_docs, generator=g, device=dev)[: args.docs] for _ in range(args.reps + 3)]
    sets = [(starts_all[p].contiguous(), lens_all[p].to(torch.int32).contiguous()) for p in picks]
    if args.batch:
        nq = args.batch
        qs = torch.randn((nq * args.lq, args.hidden), generator=g, device=dev, dtype=torch.float32).to(tdt)
        q_off = [j * args.lq for j in range(nq + 1)]
        c_off = [j * args.docs for j in range(nq + 1)]
        bsets = []
        for rep in range(6):
            pk = torch.cat([torch.randperm(args.store_docs, generator=g, device=dev)[: args.docs] for _ in range(nq)])
            bsets.append((starts_all[pk].contiguous(), lens_all[pk].to(torch.int32).contiguous()))
        out = maxsim_indexed_batch(qs, q_off, store, bsets[0][0], bsets[0][1], c_off, mode=args.mode)
        if not args.no_check:
            one = torch.cat([maxsim_indexed(qs[q_off[j]:q_off[j + 1]], store, bsets[0][0][c_off[j]:c_off[j + 1]],
                                            bsets[0][1][c_off[j]:c_off[j + 1]], mode=args.mode) for j in range(nq)])
            assert torch.equal(out, one), float((out - one).abs().max())
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
        for (e0, e1), (s, l) in zip(evs, bsets[1:]):
            e0.record()
            out = maxsim_indexed_batch(qs, q_off, store, s, l, c_off, mode=args.mode)
            e1.record()
        torch.cuda.synchronize()
        ms = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
        byts = [(int(l.sum().item()) + nq * args.lq) * args.hidden * esize for _, l in bsets[1:]]
        mean_ms, mean_bytes = sum(ms) / len(ms), sum(byts) / len(byts)
        print(json.dumps({"what": "ts_maxsim_indexed_batch, one launch", "queries": nq, "docs_per_query": args.docs,
                          "lq": args.lq, "hidden": args.hidden, "dtype": args.dtype,
                          "ms_mean": round(mean_ms, 4), "ms_min": round(ms[0], 4), "us_per_query": round(mean_ms * 1e3 / nq, 2),
                          "algorithmic_MB": round(mean_bytes / 1e6, 1), "GBps_mean": round(mean_bytes / mean_ms / 1e6, 1),
                          "frac_of_8TBps": round(mean_bytes / mean_ms / 1e6 / 8000, 4),
                          "equals_per_query_calls": None if args.no_check else True}))
        return
    for s, l in sets[:3]:
        out = maxsim_indexed(q, store, s, l, mode=args.mode)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.reps)]
    for (e0, e1), (s, l) in zip(evs, sets[3:]):
        e0.record()
        out = maxsim_indexed(q, store, s, l, mode=args.mode)
        e1.record()
    torch.cuda.synchronize()
    ms = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
    byts = [(int(l.sum().item()) + args.lq) * args.hidden * esize for _, l in sets[3:]]
    mean_ms = sum(ms) / len(ms)
    mean_bytes = sum(byts) / len(byts)
    print(json.dumps({"what": "ts_maxsim_indexed, all launches of one query", "docs": args.docs, "lq": args.lq,
                      "hidden": args.hidden, "dtype": args.dtype, "store_GB": round(rows * args.hidden * esize / 1e9, 2),
                      "ms_mean": round(mean_ms, 4), "ms_median": round(ms[len(ms) // 2], 4), "ms_min": round(ms[0], 4),
                      "algorithmic_MB": round(mean_bytes / 1e6, 1),
                      "GBps_mean": round(mean_bytes / mean_ms / 1e6, 1), "GBps_best": round(mean_bytes / ms[0] / 1e6, 1),
                      "frac_of_8TBps": round(mean_bytes / mean_ms / 1e6 / 8000, 4), "checksum": float(out.sum().item())}))


if __name__ == "__main__":
    main()
