#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03c; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_pipeline_gpu.py tests/test_configs_gpu.py tests/test_adapters_gpu.py tests/test_sharded_gpu.py -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout -k 10 300 python bench_pipeline.py --queries 64 --store --ids > $O/pq_eager.json 2> $O/pq_eager.err
timeout -k 10 300 python bench_pipeline.py --queries 64 --store --ids --graphs --cprofile > $O/pq_graphs.json 2> $O/pq_graphs.prof
timeout -k 10 300 python bench_pipeline.py --queries 256 --store --ids --many 64 > $O/many.json 2> $O/many.err
for f in pq_eager pq_graphs many; do tail -1 $O/$f.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f', d['value'], d['mean_stage_seconds'])"; done
bash tools/trace_maxsim.sh 1000 > $O/ms_trace.txt 2>&1
cat $O/ms_trace.txt
