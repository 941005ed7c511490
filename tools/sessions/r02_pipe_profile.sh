#!/bin/bash
# rocprofv3 kernel stats of the array-path pipeline (search_many, 256 queries): where the GPU time of stages 1-3 goes
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_pipe_prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench_pipeline.py --queries 256 --store --many 64 --ids > $O/bench.log 2>$O/err.log
echo "rc=$?"
f=$(find $O/trace -name "*_kernel_stats.csv" | head -1)
python3 - "$f" > $O/top.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total GPU kernel time ms", round(tot / 1e6, 2))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:40]:
    print(f'{float(r["TotalDurationNs"])/1e6:9.2f} ms {float(r["Percentage"]):6.2f}% calls {r["Calls"]:>6} avg {float(r["AverageNs"])/1e3:9.1f} us  {r["Name"][:110]}')
PY
cat $O/top.txt | head -45; tail -1 $O/bench.log | cut -c1-200
rm -rf $O/trace/*/*_kernel_trace.csv
