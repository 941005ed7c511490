#!/bin/bash
# builds experiment variants of the fused FFN kernel next to the product sources and times them (tuning only)
set -e
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT/tristage-rag_amd/csrc
SRCS="ts_index.hip ts_scan.hip ts_scan_f32s.hip ts_fused.hip ts_select.hip ts_maxsim.hip ts_maxsim16.hip ts_bm25.hip ts_fwd.hip"
E=$ROOT/tools/experiments
build() { /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I. $3 -shared -o $E/ffn_$1.so $SRCS $E/$2; }
build base ts_ffn.hip "" &
build nogelu ts_ffn.hip "-DFF_NO_GELU" &
build noloop ts_ffn.hip "-DFF_NO_LOOP" &
wait
cd $ROOT
for v in base nogelu noloop; do
  echo "== $v"; FFN_LIB=$E/ffn_$v.so FFN_ONE=${FFN_ONE:-1} timeout -k 10 120 python tools/experiments/ffn_probe.py 2>&1 | tail -3 | cut -c1-240
done
rm -f $E/ffn_*.so
