#!/bin/bash
O=gpurun_out/r02_attn; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_pipeline_gpu.py -x -q -m gpu -k "layernorm or written" 2>&1 | tail -3 || exit 1
S3_ONLY_FIRST=1 timeout -k 10 300 python3 tools/s3_forward_probe.py
