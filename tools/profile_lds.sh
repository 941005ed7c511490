#!/bin/bash
# LDS bank-conflict / MFMA counters of the two streaming kernels (own rocprofv3 passes, SQ block only).
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_lds
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_LDS_[A-Z_0-9]*\|SQ_INSTS_LDS\|SQ_ACTIVE_INST_LDS\|SQ_WAIT_INST_LDS" | sort -u | tr '\n' ' ' > $OUT/lds_counters_available.txt
C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE"
rocprofv3 --pmc $C --output-format csv -d $OUT/scan -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/scan.log 2>&1
echo "scan rc=$?"
rocprofv3 --pmc $C --output-format csv -d $OUT/maxsim -- python3 $R/tools/bench_maxsim.py --batch 64 --no-check > $OUT/maxsim.log 2>&1
echo "maxsim rc=$?"
python3 - $OUT <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
res = {}
for name, pat in (("scan", "scan_kernel<1, 2, 1>"), ("maxsim", "maxsim16_kernel")):
    fs = glob.glob(out + f"/{name}/*/*_counter_collection.csv")
    if not fs:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if pat in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    d = {k: sum(v) / len(v) for k, v in agg.items()}
    if d:
        d["launches"] = len(next(iter(agg.values())))
        if d.get("SQ_LDS_IDX_ACTIVE"):
            d["lds_bank_conflict_fraction"] = round(d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"], 5)
        if d.get("GRBM_GUI_ACTIVE"):
            d["mfma_utilisation"] = round(d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (d["GRBM_GUI_ACTIVE"] / 8 * 1024), 4)
        if d.get("SQ_WAVE_CYCLES"):
            d["wave_parked_fraction"] = round(d.get("SQ_WAIT_ANY", 0.0) / d["SQ_WAVE_CYCLES"], 4)
    res[name] = d
print(json.dumps(res))
open(out + "/summary.json", "w").write(json.dumps(res, indent=1))
PY
cat $OUT/lds_counters_available.txt
