#!/bin/bash
# ablations of ts_linear_act: no stores / no MFMA
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03p; mkdir -p $O; cd $R/tristage-rag_amd/csrc
build() { # name, flags
  mkdir -p /tmp/fsv_$1
  for f in ts_index.hip ts_scan.hip ts_scan_f32s.hip ts_fused.hip ts_select.hip ts_maxsim.hip ts_maxsim16.hip ts_bm25.hip ts_linear.hip; do /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DTS_TUNING $2 -c $f -o /tmp/fsv_$1/${f%.hip}.o 2>/dev/null & done
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DTS_TUNING -mllvm -amdgpu-mfma-vgpr-form -c ts_fwd.hip -o /tmp/fsv_$1/ts_fwd.o 2>/dev/null &
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../variants_$1.so /tmp/fsv_$1/*.o
}
build base ""
build nostore "-DFS_DBG_NOSTORE"
build linestore "-DFS_DBG_LINESTORE"
cd $R
for v in base nostore linestore; do
  echo "== $v" | tee -a $O/abl.txt
  TRISTAGE_LIB=$R/tristage-rag_amd/variants_$v.so timeout -k 10 200 python tools/linear_probe.py 2>>$O/probe.err | tail -1 | tee -a $O/abl.txt
done
rm -f tristage-rag_amd/variants_*.so
