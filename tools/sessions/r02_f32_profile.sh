#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02h
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
A="--rows 10000000 --dtype f32 --no-cpu-baseline --no-encode-leg"
python3 $R/bench.py $A --steps 10 --warmup 2 > $O/bench_f32.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py $A --steps 10 --warmup 2 > $O/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py $A --steps 3 --warmup 1 > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py $A --steps 3 --warmup 1 > $O/write.log 2>&1
cd $R
python3 - $O <<'PY'
import csv, glob, sys, json, collections
o=sys.argv[1]
res={}
for r in csv.DictReader(open(glob.glob(o+"/trace/*/*kernel_stats.csv")[0])):
    if "scan_f32s_kernel<1>" in r["Name"] or "scan_f32s_kernelILi1" in r["Name"]:
        res["kernel"]=r["Name"]; res["calls"]=int(r["Calls"]); res["avg_us"]=round(float(r["AverageNs"])/1e3,1); res["min_us"]=round(float(r["MinNs"])/1e3,1)
for name,key in (("pmc_fetch","FETCH_SIZE"),("pmc_write","WRITE_SIZE")):
    fs=glob.glob(o+f"/{name}/*/*_counter_collection.csv")
    if fs:
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(fs[0])):
            if "scan_f32s_kernel" in r["Kernel_Name"] and r["Counter_Name"]==key: agg[r["Grid_Size"]].append(float(r["Counter_Value"]))
        big=max(agg.items(), key=lambda kv: sum(kv[1])/len(kv[1])) if agg else None
        if big: res[key+"_KB_avg_per_launch"]=round(sum(big[1])/len(big[1]),1); res[key+"_launches"]=len(big[1])
if "FETCH_SIZE_KB_avg_per_launch" in res:
    res["hbm_bytes_per_launch_corrected"]=round(2*res["FETCH_SIZE_KB_avg_per_launch"]*1024+res.get("WRITE_SIZE_KB_avg_per_launch",0)*1024)
    res["algorithmic_bytes_per_launch"]=10_000_000*768*4
res["bench_line"]=[l for l in open(o+"/bench_f32.json") if l.startswith("{")][-1].strip()
json.dump(res, open(o+"/r02_f32_split_scan.json","w"), indent=1)
print(json.dumps({k:v for k,v in res.items() if k!="bench_line"}, indent=1))
PY
