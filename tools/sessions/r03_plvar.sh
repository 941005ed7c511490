#!/bin/bash
# ablations of ts_linear_add_layernorm (wrong results except x_nt): where does a chunk phase's time go?
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03p; mkdir -p $O
cd $R/tristage-rag_amd/csrc
mkdir -p /tmp/plv
for f in ts_index.hip ts_scan.hip ts_scan_f32s.hip ts_fused.hip ts_select.hip ts_maxsim.hip ts_maxsim16.hip ts_bm25.hip; do /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DTS_TUNING -c $f -o /tmp/plv/${f%.hip}.o 2>/dev/null & done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DTS_TUNING -mllvm -amdgpu-mfma-vgpr-form -c ts_fwd.hip -o /tmp/plv/ts_fwd.o 2>/dev/null &
wait
cd $R
for v in BASE DBG_X_NT DBG_NO_REFILL DBG_NO_MFMA "DBG_NO_MFMA -DDBG_NO_REFILL"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DTS_TUNING -D$v -I include -c tristage-rag_amd/csrc/ts_linear.hip -o /tmp/plv/ts_linear.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o tristage-rag_amd/variants_plv.so /tmp/plv/*.o
  echo "variant $v:"; TRISTAGE_LIB=$R/tristage-rag_amd/variants_plv.so timeout -k 10 120 python tools/proj_ln_probe.py 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k: v['fused_ms'] for k, v in d.items() if isinstance(v, dict)})" | tee -a $O/variants.txt
done
rm -f tristage-rag_amd/variants_plv.so
