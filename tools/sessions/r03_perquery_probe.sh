#!/bin/bash
# round 3: where does search() (one query, array path) spend its time? cProfile of the host side with and without
# HIP graphs, the single-query MaxSim timeline, random vs contiguous candidates.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03b; mkdir -p $O; cd $R
timeout -k 10 300 python bench_pipeline.py --queries 64 --store --ids --cprofile > $O/pq_eager.json 2> $O/pq_eager.prof
timeout -k 10 300 python bench_pipeline.py --queries 64 --store --ids --graphs --cprofile > $O/pq_graphs.json 2> $O/pq_graphs.prof
tail -1 $O/pq_eager.json | cut -c1-400; tail -1 $O/pq_graphs.json | cut -c1-400
timeout -k 10 200 python tools/bench_maxsim.py > $O/ms_random.json 2>&1
timeout -k 10 200 python tools/bench_maxsim.py --contiguous > $O/ms_contig.json 2>&1
tail -1 $O/ms_random.json; tail -1 $O/ms_contig.json
bash tools/trace_maxsim.sh 1000 > $O/ms_trace.txt 2>&1
cat $O/ms_trace.txt
