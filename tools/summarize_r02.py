#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/sessions/r02_final_a.sh into the small files kept under profiles/."""
import collections, csv, glob, json, os, sys
src = sys.argv[1]
OURS = ("scan_kernel", "fused_kernel", "select_kernel", "tau_kernel", "qprep", "relayout", "head_start", "maxsim")

def kernel_stats(d, out):
    fs = glob.glob(os.path.join(src, d, "*", "*_kernel_stats.csv"))
    if not fs:
        return
    rows = [r for r in csv.DictReader(open(fs[0]))]
    keep = [r for r in rows if any(k in r["Name"] for k in OURS)]
    other_ns = sum(float(r["TotalDurationNs"]) for r in rows if r not in keep)
    with open(os.path.join(src, out), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "StdDev"])
        for r in keep:
            w.writerow([r["Name"][:100], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["StdDev"]])
        w.writerow(["(all other kernels: torch data generation / RNG / copies)", len(rows) - len(keep), int(other_ns), "", "", "", ""])
    print(open(os.path.join(src, out)).read())

def pmc(dfetch, dwrite, out):
    agg = collections.defaultdict(list)
    for d in (dfetch, dwrite):
        fs = glob.glob(os.path.join(src, d, "*", "*_counter_collection.csv"))
        if not fs:
            continue
        for r in csv.DictReader(open(fs[0])):
            if any(k in r["Kernel_Name"] for k in OURS):
                agg[(r["Kernel_Name"][:100], r["Counter_Name"], r["VGPR_Count"], r["LDS_Block_Size"], r["Workgroup_Size"], r["Grid_Size"])].append(float(r["Counter_Value"]))
    rows = [{"kernel": k, "counter": c, "dispatches": len(v), "avg_value_KB": round(sum(v) / len(v), 1), "vgpr": vg, "lds_bytes": lds,
             "workgroup": wg, "grid_threads": grid} for (k, c, vg, lds, wg, grid), v in sorted(agg.items())]
    if rows:
        with open(os.path.join(src, out), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(rows)
        print(open(os.path.join(src, out)).read())
    return rows

def traffic_of(rows, needle):
    t = {}
    for r in rows:
        if needle in r["kernel"] and r["dispatches"] >= 3:
            t[r["counter"]] = max(t.get(r["counter"], 0.0), r["avg_value_KB"])
    if "FETCH_SIZE" in t:
        return {"bytes_per_launch": round(2 * t["FETCH_SIZE"] * 1024 + t.get("WRITE_SIZE", 0.0) * 1024), "FETCH_SIZE_KB_raw": t["FETCH_SIZE"],
                "WRITE_SIZE_KB_raw": t.get("WRITE_SIZE"),
                "correction": "2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950: FETCH_SIZE counts 64 B per 128-B request; MI355X_MICROARCH.md)"}
    return None

kernel_stats("trace", "r02_kernel_stats.csv")
kernel_stats("trace_1p25", "r02_kernel_stats_1p25M_one_launch.csv")
rows = pmc("pmc_fetch", "pmc_write", "r02_pmc_summary.csv") or []
rows2 = pmc("pmc_fetch_1p25", "pmc_write_1p25", "r02_pmc_summary_1p25M_one_launch.csv") or []
out = {"10000000x768xf16": traffic_of(rows, "scan_kernel<1, 2, 1>"), "1250000x768xf16_one_launch": traffic_of(rows2, "fused_kernel")}
json.dump(out, open(os.path.join(src, "r02_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
