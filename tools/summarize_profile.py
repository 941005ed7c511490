#!/usr/bin/env python3
"""Condense gpurun_out/prof_rNN (rocprofv3 output) into the small files kept under profiles/.

usage: tools/summarize_profile.py gpurun_out/prof_r01 r01 10000000x768xf16
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, tag, key = sys.argv[1], sys.argv[2], sys.argv[3]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")
os.makedirs(out, exist_ok=True)

newest = lambda pattern: sorted(glob.glob(pattern), key=os.path.getmtime)[-1:]   # (gpurun merges every session's files into gpurun_out/: the last one)
stats = newest(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join(out, f"{tag}_kernel_stats.csv"))

rows = []
traffic = {}
for name in ("pmc_fetch", "pmc_write"):
    files = newest(os.path.join(src, name, "*", "*_counter_collection.csv"))
    if not files:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        agg[(r["Kernel_Name"], r["Counter_Name"], r["VGPR_Count"], r["LDS_Block_Size"],
             r["Workgroup_Size"], r["Grid_Size"])].append(float(r["Counter_Value"]))
    for (k, c, vg, lds, wg, grid), v in agg.items():
        if "scan_kernel" in k or "select_kernel" in k or "tau_kernel" in k or "qprep" in k or "relayout" in k:
            avg = sum(v) / len(v)
            rows.append({"kernel": k[:80], "counter": c, "dispatches": len(v), "avg_value_KB": round(avg, 1),
                         "vgpr": vg, "lds_bytes": lds, "workgroup": wg, "grid_threads": grid})
            if "scan_kernel<1, 2, 1>" in k or "scan_kernel<2, 2, 1>" in k:
                traffic[c] = avg
with open(os.path.join(out, f"{tag}_pmc_summary.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader()
    w.writerows(rows)

# HBM bytes per launch of the fused scan+filter kernel, corrected as
# MI355X_MICROARCH.md §HBM prescribes: FETCH_SIZE (KB) reads exactly half of a wide
# coalesced stream on gfx950 -> x2; WRITE_SIZE is exact.
tpath = os.path.join(out, "traffic.json")
t = json.load(open(tpath)) if os.path.exists(tpath) else {}
if "FETCH_SIZE" in traffic:
    t[key] = round(2 * traffic["FETCH_SIZE"] * 1024 + traffic.get("WRITE_SIZE", 0.0) * 1024)
    t[key + "_detail"] = {"FETCH_SIZE_KB_raw": traffic["FETCH_SIZE"], "WRITE_SIZE_KB_raw": traffic.get("WRITE_SIZE"),
                          "correction": "2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950: FETCH_SIZE counts 64 B per 128-B request)",
                          "source": f"profiles/{tag}_pmc_summary.csv"}
json.dump(t, open(tpath, "w"), indent=1, sort_keys=True)
print(open(os.path.join(out, f"{tag}_pmc_summary.csv")).read())
print(json.dumps(t, indent=1))
