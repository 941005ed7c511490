#!/bin/bash
# bench.py on the per-GPU shapes of the other BASELINE.json configs (documentation runs)
mkdir -p gpurun_out
run() { echo "== $*" >> gpurun_out/shapes.log; timeout -k 10 200 python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 >> gpurun_out/shapes.log; }
rm -f gpurun_out/shapes.log
run --rows 1250000 --steps 100 --warmup 5                       # cfg3 shard at 8 GPUs
run --rows 2500000 --steps 60 --warmup 5                        # cfg3 shard at 4 GPUs
run --rows 5000000 --steps 40 --warmup 5                        # cfg3 shard at 2 GPUs
run --rows 6250000 --dim 1024 --dtype bf16 --steps 30 --warmup 3   # cfg4 shard at 8 GPUs
run --rows 5183 --dim 384 --dtype f32 --k 100 --steps 200 --warmup 10   # cfg1 (SciFact shape)
run --rows 5183 --dim 384 --dtype f32 --k 100 --steps 200 --warmup 10 --sync
run --rows 10000000 --batch 1 --steps 20 --warmup 3 --sync      # one query at a time (reference behaviour)
cat gpurun_out/shapes.log
