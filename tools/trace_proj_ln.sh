#!/bin/bash
# builds a traced library (-DTS_TUNING -DPL_TRACE) and prints the per-wave timeline of ts_linear_add_layernorm
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/tristage-rag_amd/csrc
mkdir -p /tmp/pltr
for f in ts_index.hip ts_scan.hip ts_scan_f32s.hip ts_fused.hip ts_select.hip ts_maxsim.hip ts_maxsim16.hip ts_bm25.hip ts_linear.hip; do /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DTS_TUNING -DPL_TRACE -c $f -o /tmp/pltr/${f%.hip}.o 2>/dev/null & done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DTS_TUNING -mllvm -amdgpu-mfma-vgpr-form -c ts_fwd.hip -o /tmp/pltr/ts_fwd.o 2>/dev/null &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../variants_pltrace.so /tmp/pltr/*.o
cd $R
for shape in "384 384" "384 1536"; do
  TRISTAGE_LIB=$R/tristage-rag_amd/variants_pltrace.so timeout -k 10 120 python tools/trace_proj_ln.py $shape 2>&1 | tail -24
done
rm -f tristage-rag_amd/variants_pltrace.so
