#!/bin/bash
# builds a traced library (-DTS_TUNING -DFS_TRACE: the FS_STAMP time stamps of ffn_stream_kernel) and prints the per-wave
# timeline of the streamed-weight projection
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/tristage-rag_amd/csrc
mkdir -p /tmp/fstr
for f in ts_index.hip ts_scan.hip ts_scan_f32s.hip ts_fused.hip ts_select.hip ts_maxsim.hip ts_maxsim16.hip ts_bm25.hip ts_linear.hip; do /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DTS_TUNING -DFS_TRACE -fno-slp-vectorize -c $f -o /tmp/fstr/${f%.hip}.o 2>/dev/null & done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DTS_TUNING -mllvm -amdgpu-mfma-vgpr-form -c ts_fwd.hip -o /tmp/fstr/ts_fwd.o 2>/dev/null &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../variants_fstrace.so /tmp/fstr/*.o
cd $R
for shape in "1152 384 0" "1536 384 1"; do
  TRISTAGE_LIB=$R/tristage-rag_amd/variants_fstrace.so timeout -k 10 120 python tools/trace_linear.py $shape 2>&1 | tail -18
done
rm -f tristage-rag_amd/variants_fstrace.so
