#!/usr/bin/env python3
"""How much host time do allocations cost inside search_many?  Wraps torch.empty with a timer around bench_pipeline's timed
loop and prints the caching allocator's segment counters before / after (device allocations inside the loop = hipMalloc)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_pipeline

acc = {"n": 0, "s": 0.0, "big_n": 0, "big_s": 0.0}
_empty = torch.empty


def timed_empty(*a, **k):
    t0 = time.perf_counter()
    out = _empty(*a, **k)
    dt = time.perf_counter() - t0
    acc["n"] += 1; acc["s"] += dt
    if dt > 50e-6:
        acc["big_n"] += 1; acc["big_s"] += dt
    return out


args = bench_pipeline.parse_args(["--queries", "256", "--store", "--many", "64", "--ids"] + sys.argv[1:])
r0 = bench_pipeline.run(args)          # warms everything (allocator included)
st0 = torch.cuda.memory_stats()
torch.empty = timed_empty
t0 = time.perf_counter()
r1 = bench_pipeline.run(args)
wall = time.perf_counter() - t0
torch.empty = _empty
st1 = torch.cuda.memory_stats()
keys = ("num_device_alloc", "num_device_free", "num_alloc_retries", "segment.all.allocated", "reserved_bytes.all.peak")
print(json.dumps({"qps_first": r0["value"], "qps_second": r1["value"], "torch_empty_calls": acc["n"], "torch_empty_seconds": round(acc["s"], 4),
                  "calls_over_50us": acc["big_n"], "seconds_in_those": round(acc["big_s"], 4), "run_wall_s": round(wall, 2),
                  "allocator": {k: (st0.get(k), st1.get(k)) for k in keys}}))
