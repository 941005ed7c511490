#!/bin/bash
# A/B of the single-query MaxSim slicing: equal tile counts on fewer waves vs T / n_waves each (tuning build, TS_M16_NO_EQ)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03y; mkdir -p $O; cd $R/tristage-rag_amd/csrc
mkdir -p /tmp/m16ab
for f in ts_index.hip ts_scan.hip ts_scan_f32s.hip ts_fused.hip ts_select.hip ts_maxsim.hip ts_maxsim16.hip ts_bm25.hip ts_linear.hip; do /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DTS_TUNING -c $f -o /tmp/m16ab/${f%.hip}.o 2>/dev/null & done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DTS_TUNING -mllvm -amdgpu-mfma-vgpr-form -c ts_fwd.hip -o /tmp/m16ab/ts_fwd.o 2>/dev/null &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../variants_m16ab.so /tmp/m16ab/*.o
cd $R
: > $O/ab.txt
for rep in 1 2; do
for n in 600 1000 1500 2000 4000; do
  for mode in eq old; do
    if [ $mode = old ]; then export TS_M16_NO_EQ=1; else unset TS_M16_NO_EQ; fi
    TRISTAGE_LIB=$R/tristage-rag_amd/variants_m16ab.so timeout -k 10 200 python tools/bench_maxsim.py --docs $n --reps 100 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode docs', d['docs'], 'ms_mean', d['ms_mean'], 'median', d['ms_median'], 'min', d['ms_min'])" | tee -a $O/ab.txt
  done
done
done
unset TS_M16_NO_EQ
rm -f tristage-rag_amd/variants_m16ab.so
