#!/bin/bash
# ablation: is the streamed-weight projection bound by the LDS reads of its B operands?  -DDBG_ONE_B reads one operand per k group
# instead of QH (wrong results, a third of the LDS traffic)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03q; mkdir -p $O
cd $R/tristage-rag_amd/csrc
mkdir -p /tmp/oneb
for f in ts_index.hip ts_scan.hip ts_scan_f32s.hip ts_fused.hip ts_select.hip ts_maxsim.hip ts_maxsim16.hip ts_bm25.hip ts_linear.hip; do /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DTS_TUNING -DDBG_ONE_B -c $f -o /tmp/oneb/${f%.hip}.o 2>/dev/null & done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DTS_TUNING -mllvm -amdgpu-mfma-vgpr-form -c ts_fwd.hip -o /tmp/oneb/ts_fwd.o 2>/dev/null &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../variants_oneb.so /tmp/oneb/*.o
cd $R
echo "product:"; timeout -k 10 120 python tools/linear_probe.py | tee $O/linear_product.json; timeout -k 10 120 python tools/proj_ln_probe.py | tee $O/projln_product.json
echo "one B operand per k group:"; TRISTAGE_LIB=$R/tristage-rag_amd/variants_oneb.so timeout -k 10 120 python tools/linear_probe.py | tee $O/linear_oneb.json
TRISTAGE_LIB=$R/tristage-rag_amd/variants_oneb.so timeout -k 10 120 python tools/proj_ln_probe.py | tee $O/projln_oneb.json
rm -f tristage-rag_amd/variants_oneb.so
