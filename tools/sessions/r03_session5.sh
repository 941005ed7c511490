#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03q; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_sharded_gpu.py tests/test_bm25_gpu.py tests/test_bench_launcher.py -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout -k 10 200 python tools/write_probe.py 2>/dev/null | tail -1 | tee $O/write_probe.json
