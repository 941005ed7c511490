#!/bin/bash
# rocprofv3 kernel trace of the stage-2 MaxSim bench (per-kernel durations for profiles/).
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_maxsim
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/bench_maxsim.py "$@" > $OUT/trace_bench.log 2>&1
echo "trace rc=$?"
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if "maxsim" in r["Name"]:
        print(f'{r["Name"][:70]:70s} calls={r["Calls"]:>5s} avg_us={float(r["AverageNs"])/1e3:8.2f} min_us={float(r["MinNs"])/1e3:8.2f}')
PY
