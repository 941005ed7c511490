#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03k; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_pipeline_gpu.py -x -q -k "tiled_linear or packed or array_path or lean" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout -k 10 200 python tools/linear_probe.py 2>$O/probe.err | tail -1 | tee $O/probe.json
bash tools/trace_linear.sh > $O/trace.txt 2>&1; grep -v amdgpu.ids $O/trace.txt
S3_ONLY_FIRST=1 timeout -k 10 200 python tools/s3_forward_probe.py 2>/dev/null | tail -1 | tee $O/s3.json
