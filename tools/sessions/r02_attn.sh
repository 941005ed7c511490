#!/bin/bash
# attention kernel: parity tests, then the array-path pipeline with and without it
O=gpurun_out/r02_attn; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_pipeline_gpu.py -x -q -m gpu -k "attention or array_path or layernorm" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
for extra in "" "--torch-attention"; do
  timeout -k 10 300 python bench_pipeline.py --queries 256 --store --many 64 --ids $extra 2>>$O/err.log | tail -1 >> $O/pipeline.jsonl
done
cat $O/pipeline.jsonl
