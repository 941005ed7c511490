#!/bin/bash
# ablations of ts_mlp_add_layernorm (wrong results): what a down phase with the activation inside costs over one without
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03j; mkdir -p $O
cd $R/tristage-rag_amd/csrc
mkdir -p /tmp/mla
for f in ts_index.hip ts_scan.hip ts_scan_f32s.hip ts_fused.hip ts_select.hip ts_maxsim.hip ts_maxsim16.hip ts_bm25.hip; do /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DTS_TUNING -c $f -o /tmp/mla/${f%.hip}.o 2>/dev/null & done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DTS_TUNING -fno-slp-vectorize -c ts_linear.hip -o /tmp/mla/ts_linear.o 2>/dev/null &
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DTS_TUNING -mllvm -amdgpu-mfma-vgpr-form -c ts_fwd.hip -o /tmp/mla/ts_fwd.o 2>/dev/null &
wait
cd $R
for v in BASE DBG_ML_TAB0 DBG_ML_NOGELU; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DTS_TUNING -D$v -fno-slp-vectorize -I include -c tristage-rag_amd/csrc/ts_mlp.hip -o /tmp/mla/ts_mlp.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o tristage-rag_amd/variants_mla.so /tmp/mla/*.o
  echo -n "variant $v: "; TRISTAGE_LIB=$R/tristage-rag_amd/variants_mla.so timeout -k 10 120 python tools/proj_ln_probe.py 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['feed_forward_block']['one_kernel_ms'])" | tee -a $O/variants.txt
done
rm -f tristage-rag_amd/variants_mla.so
