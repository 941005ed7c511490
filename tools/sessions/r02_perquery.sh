#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02d; mkdir -p $O; cd $R
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/gpu_full.log 2>&1; tail -4 $O/gpu_full.log
: > $O/pipeline.jsonl
timeout -k 10 300 python bench_pipeline.py --queries 64 --store 2>>$O/err.log | tail -1 >> $O/pipeline.jsonl
timeout -k 10 300 python bench_pipeline.py --queries 64 --store --graphs 2>>$O/err.log | tail -1 >> $O/pipeline.jsonl
timeout -k 10 300 python bench_pipeline.py --queries 64 --store --ids 2>>$O/err.log | tail -1 >> $O/pipeline.jsonl
timeout -k 10 300 python bench_pipeline.py --queries 64 --store --ids --graphs 2>>$O/err.log | tail -1 >> $O/pipeline.jsonl
python - <<'PY'
import json
for l in open("gpurun_out/r02d/pipeline.jsonl"):
    r = json.loads(l); c = r["config"]
    print(round(r["value"], 1), {k: c[k] for k in ("queries_per_search_many", "stage3_token_id_cache", "array_path", "hip_graphs") if k in c}, r["mean_stage_seconds"], r.get("index_build_s"))
PY
