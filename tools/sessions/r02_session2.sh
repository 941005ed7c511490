#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02b
mkdir -p $O
cd $R
timeout -k 10 500 python -m pytest tests/test_stage1_gpu.py tests/test_pipeline_gpu.py tests/test_stress_gpu.py -m gpu -q -x --timeout 300 > $O/tests.log 2>&1
tail -3 $O/tests.log
b() { timeout -k 10 300 python bench.py --no-cpu-baseline --no-encode-leg "$@" 2>>$O/err.log | tail -1; }
: > $O/shapes.jsonl
b --rows 1250000 --steps 200 --warmup 10 --force-exchange --pipeline on >> $O/shapes.jsonl
b --rows 1250000 --steps 200 --warmup 10 --force-exchange --pipeline on --classic >> $O/shapes.jsonl
b --rows 1250000 --steps 200 --warmup 10 >> $O/shapes.jsonl
b --rows 1250000 --steps 200 --warmup 10 --classic >> $O/shapes.jsonl
b --rows 1250000 --steps 100 --warmup 10 --sync >> $O/shapes.jsonl
b --rows 1250000 --steps 100 --warmup 10 --sync --classic >> $O/shapes.jsonl
b --rows 2500000 --steps 80 --warmup 10 --force-exchange --pipeline on >> $O/shapes.jsonl
b --rows 5000000 --steps 40 --warmup 10 --force-exchange --pipeline on >> $O/shapes.jsonl
b --rows 10000000 --steps 20 --warmup 3 >> $O/shapes.jsonl
b --rows 10000000 --steps 20 --warmup 3 --classic >> $O/shapes.jsonl
echo shapes done
for extra in "" "--no-lean" "--s3-batch 2048" "--s3-batch 512" "--bm25"; do
  timeout -k 10 300 python bench_pipeline.py --queries 256 --store --many 64 --ids $extra 2>>$O/err.log | tail -1 >> $O/pipeline.jsonl
done
echo pipeline done
