#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03u; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_bench_launcher.py tests/test_sharded_gpu.py -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
# rehearsal (NOT a scaling measurement: both ranks share the one GPU, collectives staged through the host): the row-sharded
# pipeline on the configs[2] shape through the launcher, next to the single-process line
TS_BENCH_BACKEND=gloo timeout -k 10 400 python bench_pipeline.py --gpus 2 --queries 256 --store --ids --many 64 2> $O/shard2.err | tail -1 > $O/shard2.json
timeout -k 10 300 python bench_pipeline.py --queries 256 --store --ids --many 64 2> $O/single.err | tail -1 > $O/single.json
for f in shard2 single; do python -c "import sys,json; d=json.loads(open('$O/$f.json').read()); print('$f', d['value'], d['n_gpus'], d['mean_stage_seconds'], d.get('rank0_shard'))"; done
