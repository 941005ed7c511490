#!/bin/bash
# HBM traffic of the feed-forward block, one kernel against two: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over
# tools/proj_ln_probe.py, summarised per kernel (FETCH_SIZE doubled: the gfx950 correction of MI355X_MICROARCH.md)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_mlp
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/tools/proj_ln_probe.py > $OUT/pmc_fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/tools/proj_ln_probe.py > $OUT/pmc_write.log 2>&1; echo "write rc=$?"
python3 - $OUT <<'PY'
import csv, glob, sys, json, collections
out = sys.argv[1]
res = collections.defaultdict(dict)
for name, key in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    fs = sorted(glob.glob(out + f"/{name}/*/*_counter_collection.csv"))
    if not fs: continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[-1])):
        if r["Counter_Name"] == key and any(k in r["Kernel_Name"] for k in ("mlp_ln_kernel", "proj_ln_kernel", "ffn_stream_kernel", "add_layernorm_kernel")):
            agg[(r["Kernel_Name"][:60], r["Grid_Size"])].append(float(r["Counter_Value"]))
    for (k, grid), v in agg.items():
        res[f"{k} grid={grid}"][key + "_KB_avg"] = round(sum(v) / len(v), 1); res[f"{k} grid={grid}"]["launches"] = len(v)
for k, d in res.items():
    if "FETCH_SIZE_KB_avg" in d:
        d["HBM_MB_corrected"] = round((2 * d["FETCH_SIZE_KB_avg"] + d.get("WRITE_SIZE_KB_avg", 0)) * 1024 / 1e6, 1)
print(json.dumps(res, indent=1))
open(out + "/summary.json", "w").write(json.dumps(res, indent=1))
PY
