#!/bin/bash
# round 3: projection + residual + LayerNorm as one kernel — parity test, the probe, the stage-3 forward and the pipeline line
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03u; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_pipeline_gpu.py -x -q -k "tiled_linear or lean or cross or stage3" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout -k 10 200 python tools/proj_ln_probe.py > $O/probe.json 2> $O/probe.err; echo "probe rc=$?"; tail -1 $O/probe.json; tail -3 $O/probe.err
timeout -k 10 200 python tools/s3_forward_probe.py > $O/s3_forward.json 2> $O/s3_forward.err; echo "s3 rc=$?"; tail -2 $O/s3_forward.json
