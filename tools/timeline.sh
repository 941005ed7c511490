#!/bin/bash
# kernel timeline of the pipelined bench (gaps between consecutive scan kernels)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/timeline; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $R/bench.py --rows ${1:-1250000} --steps 40 --warmup 5 --no-cpu-baseline ${@:2} > $OUT/bench.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/t/*/*_kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(f))]
ks=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"][:40]) for r in rows]
ks.sort()
import os
pat=os.environ.get("TL_KERNEL","scan_kernel<1, 2, 1>")
scan=[k for k in ks if pat in k[2]]
last=scan[-30:]
t0=last[0][0]
gaps=[(b[0]-a[1])/1e3 for a,b in zip(last,last[1:])]
durs=[(k[1]-k[0])/1e3 for k in last]
print("scan durations us (last 30):", [round(d) for d in durs])
print("gaps between consecutive scans us:", [round(g,1) for g in gaps])
# what runs in one gap window
a,b=last[10],last[11]
print("kernels between two scans:")
b=last[13]
for k in ks:
    if k[0]>=a[0]-1000 and k[0]<=b[1]:
        print("  start %8.1f dur %7.1f  %s"%((k[0]-a[0])/1e3,(k[1]-k[0])/1e3,k[2]))
PY
