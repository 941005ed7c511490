#!/bin/bash
# A/B of MaxSim-kernel build variants on one GPU (tools/bench_maxsim.py with TRISTAGE_LIB):
# the measurements quoted in DESIGN.md 4.4 (ring depth, refill group, waves per workgroup).
set -e
mkdir -p gpurun_out
ROOT=$PWD
cd tristage-rag_amd/csrc
SRCS="ts_index.hip ts_scan.hip ts_select.hip ts_maxsim.hip ts_maxsim16.hip ts_bm25.hip"
build() { /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DTS_TUNING $2 -shared -o ../variants_$1.so $SRCS 2>/dev/null; }
build base "" &                                                    # 4 waves, ring 16, group 8
build g1 "-DM16_GROUP=1" &                                         # one slot refilled per step
build g4 "-DM16_GROUP=4" &
build w8r8 "-DM16_THREADS=512 -DM16_RING=8 -DM16_GROUP=8" &        # 8 waves x 8 KiB
build w8r16 "-DM16_THREADS=512 -DM16_RING=16 -DM16_GROUP=8" &
build r32 "-DM16_RING=32 -DM16_GROUP=8" &
build nocompute "-DM16_DBG_NOCOMPUTE" &                            # loads only
wait
cd $ROOT
run() { # name, lib, env, bench args
  env $3 TRISTAGE_LIB=$ROOT/tristage-rag_amd/variants_$2.so timeout -k 10 120 python tools/bench_maxsim.py $4 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-34s' % '$1', '%-28s' % '$4', d['ms_mean'], d['ms_min'], d['GBps_mean'])
" >> gpurun_out/variants_maxsim.log 2>&1
}
rm -f gpurun_out/variants_maxsim.log
for args in "" "--docs 2047 --len-lo 192" "--batch 64 --no-check"; do
  run "4 waves, ring 16, group 8 (product)" base X=1 "$args"
  run "group 1" g1 X=1 "$args"
  run "group 4" g4 X=1 "$args"
  run "8 waves, ring 8" w8r8 X=1 "$args"
  run "8 waves, ring 16" w8r16 X=1 "$args"
  run "4 waves, ring 32" r32 X=1 "$args"
  run "loads only (no MFMA / LDS)" nocompute X=1 "$args"
  run "2 workgroups per CU" base TS_M16_GRIDMUL=2 "$args"
done
cat gpurun_out/variants_maxsim.log
rm -f tristage-rag_amd/variants_*.so
