#!/bin/bash
# the 5.8 ms step of round 2's 4-shard rehearsal (2.5 M rows, pipelined, exchange in a world of one): where does it sit?
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03f; mkdir -p $O; cd $R
for rep in 1 2 3; do
timeout -k 10 200 python bench.py --rows 2500000 --steps 80 --warmup 5 --force-exchange --pipeline on --step-events on --dump-steps \
   --no-cpu-baseline --no-encode-leg --no-pipeline-leg --no-read-probe --traffic off 2>/dev/null | tail -1 >> $O/outlier.jsonl
done
python - <<'PY'
import json
for l in open("gpurun_out/r03f/outlier.jsonl"):
    d = json.loads(l); c = d["config"]; s = c["step_ms"]
    big = [(i, x) for i, x in enumerate(s) if x > 2 * d["ms_per_step"]]
    print(d["ms_per_step"], d["ms_per_step_min"], d["ms_per_step_max"], "finish_at", c["finish_at"], "outliers", big)
PY
