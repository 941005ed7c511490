#!/bin/bash
# the array-path pipeline lines kept under profiles/r02_bench_pipeline.jsonl
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02c; mkdir -p $O; cd $R
: > $O/pipeline.jsonl
for extra in "" "--torch-attention" "--no-lean" "--s3-batch 2048" "--bm25" "--keep"; do
  timeout -k 10 300 python bench_pipeline.py --queries 256 --store --many 64 --ids $extra 2>>$O/err.log | tail -1 >> $O/pipeline.jsonl
done
timeout -k 10 300 python bench_pipeline.py --queries 256 --store --many 64 2>>$O/err.log | tail -1 >> $O/pipeline.jsonl
timeout -k 10 300 python bench_pipeline.py --queries 64 --store 2>>$O/err.log | tail -1 >> $O/pipeline.jsonl
timeout -k 10 300 python bench_pipeline.py --queries 64 --store --graphs 2>>$O/err.log | tail -1 >> $O/pipeline.jsonl
python - <<'PY'
import json
for l in open("gpurun_out/r02c/pipeline.jsonl"):
    r = json.loads(l); c = r["config"]
    print(round(r["value"], 1), {k: c[k] for k in ("queries_per_search_many", "bm25_rrf", "stage3_token_id_cache", "stage3_pairs_per_forward", "stage3_lean_forward", "save_intermediate_results", "array_path", "hip_graphs") if k in c}, r["mean_stage_seconds"])
PY
