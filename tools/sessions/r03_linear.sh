#!/bin/bash
# ts_linear_act: rows per workgroup (QH) sweep + counters of the product configuration
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03i; mkdir -p $O; cd $R/tristage-rag_amd/csrc
SRCS="ts_index.hip ts_scan.hip ts_scan_f32s.hip ts_fused.hip ts_select.hip ts_maxsim.hip ts_maxsim16.hip ts_bm25.hip ts_linear.hip"
mkdir -p /tmp/fsobj; : > $O/build.err
for f in $SRCS; do /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DTS_TUNING -c $f -o /tmp/fsobj/${f%.hip}.o 2>> $O/build.err & done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DTS_TUNING -mllvm -amdgpu-mfma-vgpr-form -c ts_fwd.hip -o /tmp/fsobj/ts_fwd.o 2>> $O/build.err &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../variants_fs.so /tmp/fsobj/*.o 2>> $O/build.err
cd $R
: > $O/qh.jsonl
for qh in 3 4 5 6 2; do
  TS_FS_QH=$qh TRISTAGE_LIB=$R/tristage-rag_amd/variants_fs.so timeout -k 10 200 python tools/linear_probe.py 2>>$O/probe.err | tail -1 >> $O/qh.jsonl
done
cat $O/qh.jsonl
cd /tmp && export TMPDIR=/tmp
C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
rm -rf $O/pmc $O/pmc_f $O/pmc_w
rocprofv3 --pmc $C --output-format csv -d $O/pmc -- python3 $R/tools/linear_probe.py > $O/pmc.log 2>&1
echo "pmc rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python3 $R/tools/linear_probe.py > $O/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python3 $R/tools/linear_probe.py > $O/pmc_w.log 2>&1
python3 - $O <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
for sub in ("pmc", "pmc_f", "pmc_w"):
    fs = glob.glob(out + "/" + sub + "/**/*_counter_collection.csv", recursive=True)
    if not fs:
        continue
    rows = [r for r in csv.DictReader(open(fs[0])) if "ffn_stream_kernel" in r["Kernel_Name"]]
    # the probe runs qkv, attn_out, up_gelu in that order, 24 dispatches each (1 + 3 + 20)
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    third = {d: ("qkv", "attn_out", "up_gelu")[min(2, i * 3 // max(len(ids), 1))] for i, d in enumerate(ids)}
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        agg[third[int(r["Dispatch_Id"])]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(sub, k, json.dumps({c: round(sum(x) / len(x)) for c, x in v.items()}), "n=", len(next(iter(v.values()))))
PY
rm -f $R/tristage-rag_amd/variants_fs.so
