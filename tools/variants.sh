#!/bin/bash
# A/B of scan-kernel build variants on one GPU (the ablations quoted in DESIGN.md 4.1).
# Builds each variant next to the product library and runs bench.py with TRISTAGE_LIB.
set -e
mkdir -p gpurun_out
ROOT=$PWD
cd tristage-rag_amd/csrc
SRCS="ts_index.hip ts_scan.hip ts_select.hip ts_maxsim.hip ts_bm25.hip"
build() { /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DTS_TUNING $2 -shared -o ../variants_$1.so $SRCS; }
build base ""
build plain "-DTS_PLAIN_LOADS"
build ring16 "-DTS_RING=16"
build t1024 "-DSCAN_THREADS=1024"
build nomfma "-DDBG_NO_MFMA"
build nolds "-DDBG_NO_LDS"
cd $ROOT
run() { # name, lib, extra env
  env $3 TRISTAGE_LIB=$ROOT/tristage-rag_amd/variants_$2.so timeout -k 10 120 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --sync 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['roofline']['achieved'], d['config']['search_path'])
" >> gpurun_out/variants.log 2>&1
}
rm -f gpurun_out/variants.log
for round in 1 2; do
  run "base r$round" base X=1
  run "plain-loads r$round" plain X=1
  run "ring16 r$round" ring16 X=1
  run "1024-threads r$round" t1024 X=1
  run "no-mfma r$round" nomfma X=1
  run "no-lds r$round" nolds X=1
  run "tau=inf(no-survivors) r$round" base TS_DEBUG_TAU_INF=1
done
cat gpurun_out/variants.log
rm -f tristage-rag_amd/variants_*.so
