#!/bin/bash
# rocprofv3 runs of the stage-2 MaxSim bench for profiles/: kernel trace + stats, then PMC
# passes (separate runs: FETCH_SIZE, WRITE_SIZE).  usage: tools/profile_maxsim.sh [bench args]
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_maxsim
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/bench_maxsim.py "$@" > $OUT/trace_bench.log 2>&1
echo "trace rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/tools/bench_maxsim.py "$@" > $OUT/pmc_fetch.log 2>&1
echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/tools/bench_maxsim.py "$@" > $OUT/pmc_write.log 2>&1
echo "write rc=$?"
python3 - $OUT <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
res = {}
f = glob.glob(out + "/trace/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "maxsim" in r["Name"]:
        res["kernel"] = r["Name"]; res["calls"] = int(r["Calls"]); res["avg_us"] = round(float(r["AverageNs"]) / 1e3, 2)
        res["min_us"] = round(float(r["MinNs"]) / 1e3, 2)
for name, key in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    fs = glob.glob(out + f"/{name}/*/*_counter_collection.csv")
    if not fs: continue
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(fs[0])) if "maxsim16_kernel" in r["Kernel_Name"] and r["Counter_Name"] == key]
    if v: res[key + "_KB_avg_per_launch"] = round(sum(v) / len(v), 1); res[key + "_launches"] = len(v)
if "FETCH_SIZE_KB_avg_per_launch" in res:
    res["hbm_bytes_per_launch_corrected"] = round(2 * res["FETCH_SIZE_KB_avg_per_launch"] * 1024 + res.get("WRITE_SIZE_KB_avg_per_launch", 0) * 1024)
    res["correction"] = "2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950: FETCH_SIZE counts 64 B per 128-B request; MI355X_MICROARCH.md)"
res["bench_line"] = [l for l in open(out + "/trace_bench.log") if l.startswith("{")][-1].strip()
print(json.dumps(res))
open(out + "/summary.json", "w").write(json.dumps(res, indent=1))
PY
