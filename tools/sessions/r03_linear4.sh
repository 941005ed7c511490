#!/bin/bash
# stall-reason counters of ffn_stream_kernel (Q/K/V shape dominates: look at the first third of the dispatches)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03o; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*\|TCP_[A-Z_0-9]*\|TA_[A-Z_0-9]*" | sort -u | tr '\n' ' ' > $O/counters_available.txt
run() { # tag, counters
  rm -rf $O/$1
  rocprofv3 --pmc $2 --output-format csv -d $O/$1 -- python3 $R/tools/linear_probe.py > $O/$1.log 2>&1
  echo "$1 rc=$?"
}
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
run b "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM"
run c "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SALU SQ_WAVES SQ_IFETCH"
python3 - $O <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
for sub in ("a", "b", "c"):
    fs = glob.glob(out + "/" + sub + "/**/*_counter_collection.csv", recursive=True)
    if not fs:
        print(sub, "no csv"); continue
    rows = [r for r in csv.DictReader(open(fs[0])) if "ffn_stream_kernel" in r["Kernel_Name"]]
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    third = {d: ("qkv", "attn_out", "up_gelu")[min(2, i * 3 // max(len(ids), 1))] for i, d in enumerate(ids)}
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        agg[third[int(r["Dispatch_Id"])]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in ("qkv", "up_gelu"):
        print(sub, k, json.dumps({c: round(sum(x) / len(x)) for c, x in agg[k].items()}))
PY
