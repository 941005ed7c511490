#!/bin/bash
# A/B of MaxSim-kernel build variants on one GPU (tools/bench_maxsim.py with TRISTAGE_LIB).
set -e
mkdir -p gpurun_out
ROOT=$PWD
cd tristage-rag_amd/csrc
SRCS="ts_index.hip ts_scan.hip ts_select.hip ts_maxsim.hip ts_maxsim16.hip ts_bm25.hip"
build() { /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DTS_TUNING $2 -shared -o ../variants_$1.so $SRCS 2>/dev/null; }
build base "" &
build ring16 "-DM16_RING=16" &
build ring4 "-DM16_RING=4" &
build t1024 "-DM16_THREADS=1024" &
build t256 "-DM16_THREADS=256" &
wait
cd $ROOT
run() { # name, lib, env, bench args
  env $3 TRISTAGE_LIB=$ROOT/tristage-rag_amd/variants_$2.so timeout -k 10 120 python tools/bench_maxsim.py $4 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-34s' % '$1', '$4', d['ms_mean'], d['ms_min'], d['GBps_mean'])
" >> gpurun_out/variants_maxsim.log 2>&1
}
rm -f gpurun_out/variants_maxsim.log
for args in "" "--len-lo 192" "--lq 64"; do
  run "base ring8 8w grid x1" base X=1 "$args"
  run "ring8 8w grid x2" base TS_M16_GRIDMUL=2 "$args"
  run "ring8 8w grid x3" base TS_M16_GRIDMUL=3 "$args"
  run "ring16 8w grid x1" ring16 X=1 "$args"
  run "ring16 8w grid x2" ring16 TS_M16_GRIDMUL=2 "$args"
  run "ring4 8w grid x2" ring4 TS_M16_GRIDMUL=2 "$args"
  run "ring4 8w grid x3" ring4 TS_M16_GRIDMUL=3 "$args"
  run "ring8 16w grid x1" t1024 X=1 "$args"
  run "ring8 4w grid x2" t256 TS_M16_GRIDMUL=2 "$args"
  run "ring8 4w grid x4" t256 TS_M16_GRIDMUL=4 "$args"
  run "ring8 4w grid x6" t256 TS_M16_GRIDMUL=6 "$args"
done
cat gpurun_out/variants_maxsim.log
rm -f tristage-rag_amd/variants_*.so
