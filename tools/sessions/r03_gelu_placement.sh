#!/bin/bash
# A/B on one box: the erf GELU in the up projection's epilogue or in the down kernel's row staging (S3_GELU_IN_UP).  (The run kept
# in profiles/r03_s3_forward_ab_split_gelu.txt also had a build whose last partial round of tiles was a second launch with 64-row
# tiles, "split" / "no split": no difference, taken out.)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03m; mkdir -p $O
cd $R/tristage-rag_amd/csrc
mkdir -p /tmp/spl
for f in ts_index.hip ts_scan.hip ts_scan_f32s.hip ts_fused.hip ts_select.hip ts_maxsim.hip ts_maxsim16.hip ts_bm25.hip; do /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DTS_TUNING -c $f -o /tmp/spl/${f%.hip}.o 2>/dev/null & done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DTS_TUNING -fno-slp-vectorize -c ts_linear.hip -o /tmp/spl/ts_linear.o 2>/dev/null &
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DTS_TUNING -mllvm -amdgpu-mfma-vgpr-form -c ts_fwd.hip -o /tmp/spl/ts_fwd.o 2>/dev/null &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../variants_spl.so /tmp/spl/*.o
cd $R
export TRISTAGE_LIB=$R/tristage-rag_amd/variants_spl.so
for rep in 1 2; do
  echo -n "GELU in the down kernel's staging: "; S3_ONLY_FIRST=1 timeout -k 10 120 python tools/s3_forward_probe.py 2>/dev/null | tail -1 | tee -a $O/ab.txt
  echo -n "GELU in the up epilogue: "; S3_GELU_IN_UP=1 S3_ONLY_FIRST=1 timeout -k 10 120 python tools/s3_forward_probe.py 2>/dev/null | tail -1 | tee -a $O/ab.txt
done
rm -f tristage-rag_amd/variants_spl.so
