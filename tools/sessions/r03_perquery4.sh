#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03h; mkdir -p $O; cd $R
timeout -k 10 300 python bench_pipeline.py --queries 64 --store --ids > $O/pq_eager.json 2> $O/pq_eager.err
timeout -k 10 300 python bench_pipeline.py --queries 64 --store --ids --graphs --cprofile > $O/pq_graphs.json 2> $O/pq_graphs.prof
timeout -k 10 300 python bench_pipeline.py --queries 64 --store --ids --graphs > $O/pq_graphs2.json 2> $O/pq_graphs2.err
timeout -k 10 300 python bench_pipeline.py --queries 256 --store --ids --many 64 > $O/many.json 2> $O/many.err
for f in pq_eager pq_graphs pq_graphs2 many; do tail -1 $O/$f.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f', d['value'], d['mean_stage_seconds'])"; done
