#!/bin/bash
# round 3: whole -m gpu suite on the tree with the LDS-DMA query image, then the single-query and 64-query MaxSim profiles
R=${GRAFT_REPO_ROOT:-$(pwd)}
export GRAFT_REPO_ROOT=$R
O=$R/gpurun_out/r03v; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/pytest.log; tail -4 $O/pytest.log
bash tools/profile_maxsim.sh > $O/maxsim_single.log 2>&1; tail -1 $O/maxsim_single.log; cp gpurun_out/prof_maxsim/summary.json $O/r03_maxsim_single.json
cp gpurun_out/prof_maxsim/trace/*/*kernel_stats.csv $O/r03_maxsim_single_kernel_stats.csv 2>/dev/null
bash tools/profile_maxsim.sh --batch 64 --no-check > $O/maxsim_batch.log 2>&1; tail -1 $O/maxsim_batch.log; cp gpurun_out/prof_maxsim/summary.json $O/r03_maxsim_batch64.json
timeout -k 10 300 python bench_pipeline.py > $O/pipeline.json 2> $O/pipeline.err; tail -1 $O/pipeline.json | cut -c1-600
