#!/bin/bash
mkdir -p gpurun_out
run() { # name, lib, extra env
  env $3 TRISTAGE_LIB=$2 timeout -k 10 120 python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['roofline']['achieved'], d['config']['max_candidates_per_query'], d['config']['search_path'], d['config']['phase_ms_per_step'])
" >> gpurun_out/variants.log 2>&1
}
for round in 1 2; do
run "default(nt) r$round" $PWD/tristage-rag_amd/libtristage.so X=1

done
cat gpurun_out/variants.log
