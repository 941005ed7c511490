#!/bin/bash
# throughput sweep over batch size, k, rows and dtype on one GPU -> gpurun_out/sweep.jsonl
mkdir -p gpurun_out; rm -f gpurun_out/sweep.jsonl
run() { timeout -k 10 200 python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 >> gpurun_out/sweep.jsonl; }
for b in 1 8 32 64; do run --rows 10000000 --batch $b --steps 20 --warmup 3; done
for k in 10 100 2048; do run --rows 10000000 --k $k --steps 20 --warmup 3; done
for n in 100000 1000000 20000000 40000000; do run --rows $n --steps 20 --warmup 3; done
run --rows 10000000 --dim 384 --steps 20 --warmup 3
run --rows 10000000 --dim 1024 --dtype bf16 --steps 20 --warmup 3
python - <<'PY'
import json
for l in open("gpurun_out/sweep.jsonl"):
    d=json.loads(l); c=d["config"]; r=d["roofline"] or {}
    print(c["rows"], c["dim"], d["dtype"], "B", c["batch"], "k", c["k"], "->", d["value"], "q/s", d["ms_per_step"], "ms", "scan", r.get("avg_kernel_ms"), "frac", r.get("frac"), c["search_path"], c["max_candidates_per_query"])
PY
