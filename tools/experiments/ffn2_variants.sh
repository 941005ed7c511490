#!/bin/bash
set -e
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT/tristage-rag_amd/csrc
SRCS="ts_index.hip ts_scan.hip ts_scan_f32s.hip ts_fused.hip ts_select.hip ts_maxsim.hip ts_maxsim16.hip ts_bm25.hip ts_fwd.hip"
E=$ROOT/tools/experiments
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I. -shared -o $E/ffn2.so $SRCS $E/ts_ffn2.hip &
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I. -DFS_NO_STORE -shared -o $E/ffn2_nostore.so $SRCS $E/ts_ffn2.hip &
wait
cd $ROOT
for v in ffn2 ffn2_nostore; do echo "== $v"; FFN_LIB=$E/$v.so timeout -k 10 200 python tools/experiments/ffn2_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-330; done
rm -f $E/ffn2*.so
