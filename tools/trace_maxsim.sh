#!/bin/bash
# builds the traced variant of the library and prints the per-wave timeline of a single-query MaxSim launch
set -e
ROOT=$PWD
cd tristage-rag_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DTS_TUNING -DM16_TRACE -shared -o ../variants_m16trace.so ts_index.hip ts_scan.hip ts_scan_f32s.hip ts_fused.hip ts_select.hip ts_maxsim.hip ts_maxsim16.hip ts_bm25.hip ts_fwd.hip ts_linear.hip
cd $ROOT
mkdir -p gpurun_out
TRISTAGE_LIB=$ROOT/tristage-rag_amd/variants_m16trace.so timeout -k 10 200 python tools/trace_maxsim.py ${1:-1000} 2>&1 | tail -12
rm -f tristage-rag_amd/variants_m16trace.so
