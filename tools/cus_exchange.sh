#!/bin/bash
# pipelined 1.25 M-row step (exchange included) vs the number of CUs the persistent scan grid takes (tuning build)
set -e
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT/tristage-rag_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DTS_TUNING -shared -o ../variants_base.so ts_index.hip ts_scan.hip ts_scan_f32s.hip ts_fused.hip ts_select.hip ts_maxsim.hip ts_maxsim16.hip ts_bm25.hip ts_fwd.hip
cd $ROOT
O=gpurun_out/cus_exchange.log; rm -f $O
for r in 1 2; do for c in 224 208 192 176 160; do for hs in 12 0; do
  TS_SCAN_CUS=$c TS_HEAD_START_US=$hs TRISTAGE_LIB=$ROOT/tristage-rag_amd/variants_base.so timeout -k 10 120 python bench.py --rows 1250000 --steps 200 --warmup 10 --force-exchange --pipeline on --no-cpu-baseline --no-encode-leg 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cus $c head $hs r$r', d['ms_per_step'], d['config'].get('phase_ms_per_step'))" >> $O
done; done; done
cat $O; rm -f tristage-rag_amd/variants_base.so
