#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_s3_prof; mkdir -p $O
timeout -k 10 300 python3 $R/tools/s3_forward_probe.py > $O/probe.json 2>$O/err.log || { tail $O/err.log; exit 1; }
cat $O/probe.json
cd /tmp && export TMPDIR=/tmp
S3_ONLY_FIRST=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/s3_forward_probe.py > $O/prof.log 2>>$O/err.log
f=$(find $O/trace -name "*_kernel_stats.csv" | head -1)
python3 - "$f" > $O/top.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total GPU kernel time ms", round(tot / 1e6, 2), "(13 forwards)")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:25]:
    print(f'{float(r["TotalDurationNs"])/1e6/13:8.3f} ms/fwd {float(r["Percentage"]):6.2f}% calls {r["Calls"]:>5} avg {float(r["AverageNs"])/1e3:8.1f} us  {r["Name"][:100]}')
PY
cat $O/top.txt
rm -f $O/trace/*/*_kernel_trace.csv
