#!/bin/bash
# builds the traced variant of the library and prints the timeline of the one-launch search
set -e
ROOT=$PWD
cd tristage-rag_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DTS_TUNING -DFZ_TRACE -shared -o ../variants_fztrace.so ts_index.hip ts_scan.hip ts_scan_f32s.hip ts_fused.hip ts_select.hip ts_maxsim.hip ts_maxsim16.hip ts_bm25.hip ts_fwd.hip
cd $ROOT
mkdir -p gpurun_out
TRISTAGE_LIB=$ROOT/tristage-rag_amd/variants_fztrace.so timeout -k 10 200 python tools/trace_fused.py ${1:-1250000} > gpurun_out/trace_fused_${1:-1250000}.txt 2>&1 || true
cat gpurun_out/trace_fused_${1:-1250000}.txt | tail -25
echo "== sample = corpus prefix (experiment)"
TS_FUSED_PREFIX_SAMPLE=1 TRISTAGE_LIB=$ROOT/tristage-rag_amd/variants_fztrace.so timeout -k 10 200 python tools/trace_fused.py ${1:-1250000} 2>&1 | tail -16
rm -f tristage-rag_amd/variants_fztrace.so
