#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02c
mkdir -p $O
cd $R
bash tools/trace_fused.sh 1250000 2>&1 | head -21
timeout -k 10 300 python -m pytest tests/test_stage1_gpu.py -m gpu -q -x -k "one_launch" 2>&1 | tail -3
b() { timeout -k 10 300 python bench.py --no-cpu-baseline --no-encode-leg "$@" 2>>$O/err.log | tail -1; }
: > $O/shapes.jsonl
b --rows 1250000 --steps 200 --warmup 10 --sync >> $O/shapes.jsonl
b --rows 1250000 --steps 200 --warmup 10 >> $O/shapes.jsonl
b --rows 1250000 --steps 200 --warmup 10 --force-exchange --pipeline on >> $O/shapes.jsonl
b --rows 10000000 --steps 20 --warmup 3 >> $O/shapes.jsonl
python - <<'PY'
import json
for l in open("gpurun_out/r02c/shapes.jsonl"):
    if not l.startswith("{"): continue
    d=json.loads(l); c=d["config"]; r=d.get("roofline") or {}
    print("rows",c["rows"],"| ms/step",d["ms_per_step"],"| q/s",d["value"],"|", "ONE" if "one launch" in c["search_path"] else "classic","| kernel_ms",r.get("avg_kernel_ms"),"frac",r.get("frac"),"|",c["submission"][:32],"| cand",c["max_candidates_per_query"])
PY
