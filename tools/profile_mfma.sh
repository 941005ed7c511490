#!/bin/bash
# MFMA-utilisation counters of the scan kernel (own rocprofv3 pass; SQ block only).
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_mfma
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*\|GRBM_GUI_ACTIVE\|SQ_BUSY_CYCLES\|SQ_BUSY_CU_CYCLES\|SQ_WAVE_CYCLES\|SQ_WAIT_ANY\b\|SQ_ACTIVE_INST_ANY" | sort -u > $OUT/counters_available.txt
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/pmc.log 2>&1
echo "pmc rc=$?"
python3 - $OUT <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
fs = glob.glob(out + "/pmc/*/*_counter_collection.csv")
agg = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if "scan_kernel<1, 2, 1>" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: sum(v) / len(v) for k, v in agg.items()}
res["launches"] = len(next(iter(agg.values()))) if agg else 0
print(json.dumps(res))
open(out + "/summary.json", "w").write(json.dumps(res, indent=1))
PY
cat $OUT/counters_available.txt | tr '\n' ' '
