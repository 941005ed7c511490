#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03x; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_stress_gpu.py tests/test_pipeline_gpu.py tests/test_configs_gpu.py -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout -k 10 200 python tools/bench_maxsim.py > $O/ms_single.json 2>&1; tail -1 $O/ms_single.json
timeout -k 10 200 python tools/bench_maxsim.py --docs 1000 --lq 48 > $O/ms_single_lq48.json 2>&1; tail -1 $O/ms_single_lq48.json
for n in 300 600 2000 4000; do timeout -k 10 200 python tools/bench_maxsim.py --docs $n 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('docs', d['docs'], 'ms', d['ms_mean'], d['ms_min'], 'GB/s', d['GBps_mean'])"; done
timeout -k 10 200 python tools/bench_maxsim.py --batch 64 > $O/ms_batch.json 2>&1; tail -1 $O/ms_batch.json
bash tools/trace_maxsim.sh 1000 > $O/ms_trace.txt 2>&1; cat $O/ms_trace.txt
