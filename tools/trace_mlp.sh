#!/bin/bash
# builds a traced library (-DTS_TUNING -DML_TRACE) and prints the per-wave timeline of ts_mlp_add_layernorm
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/tristage-rag_amd/csrc
mkdir -p /tmp/mltr
for f in ts_index.hip ts_scan.hip ts_scan_f32s.hip ts_fused.hip ts_select.hip ts_maxsim.hip ts_maxsim16.hip ts_bm25.hip ts_linear.hip ts_mlp.hip; do /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DTS_TUNING -DML_TRACE -fno-slp-vectorize -c $f -o /tmp/mltr/${f%.hip}.o 2>/dev/null & done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DTS_TUNING -mllvm -amdgpu-mfma-vgpr-form -c ts_fwd.hip -o /tmp/mltr/ts_fwd.o 2>/dev/null &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../variants_mltrace.so /tmp/mltr/*.o
cd $R
TRISTAGE_LIB=$R/tristage-rag_amd/variants_mltrace.so timeout -k 10 120 python tools/trace_mlp.py 2>&1 | tail -12
rm -f tristage-rag_amd/variants_mltrace.so
