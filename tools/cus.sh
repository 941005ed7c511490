#!/bin/bash
# scan bandwidth vs number of CUs the persistent scan grid occupies (tuning build)
set -e
ROOT=$PWD
cd tristage-rag_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DTS_TUNING -shared -o ../variants_base.so ts_index.hip ts_scan.hip ts_select.hip ts_maxsim.hip ts_bm25.hip
cd $ROOT
rm -f gpurun_out/cus.log
for r in 1 2; do for c in 256 252 248 240 224 192; do
  TS_SCAN_CUS=$c TRISTAGE_LIB=$ROOT/tristage-rag_amd/variants_base.so timeout -k 10 120 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --sync 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cus $c r$r', d['roofline']['avg_kernel_ms'], d['roofline']['achieved'])" >> gpurun_out/cus.log
done; done
cat gpurun_out/cus.log; rm -f tristage-rag_amd/variants_base.so
