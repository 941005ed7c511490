#!/bin/bash
# round 3: the rocprofv3 summaries kept under profiles/ (kernel trace + stats of the default bench command, PMC passes,
# the single-query and batched MaxSim profiles, the stage-3 forward's kernel split)
R=${GRAFT_REPO_ROOT:-$(pwd)}
export GRAFT_REPO_ROOT=$R
O=$R/gpurun_out/r03t; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_r03; mkdir -p $R/gpurun_out/prof_r03
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03/trace -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pipeline-leg --traffic off > $O/trace_bench.log 2>&1
echo "trace rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_r03/pmc_fetch -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-pipeline-leg --no-encode-leg --traffic off > $O/pmc_fetch.log 2>&1
echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_r03/pmc_write -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-pipeline-leg --no-encode-leg --traffic off > $O/pmc_write.log 2>&1
echo "write rc=$?"
cd $R
python3 tools/summarize_profile.py gpurun_out/prof_r03 r03 10000000x768xf16 > $O/summarize.log 2>&1; tail -3 $O/summarize.log
bash tools/profile_maxsim.sh > $O/maxsim_single.log 2>&1; tail -1 $O/maxsim_single.log; cp gpurun_out/prof_maxsim/summary.json $O/r03_maxsim_single.json
cp gpurun_out/prof_maxsim/trace/*/*kernel_stats.csv $O/r03_maxsim_single_kernel_stats.csv 2>/dev/null
bash tools/profile_maxsim.sh --batch 64 --no-check > $O/maxsim_batch.log 2>&1; tail -1 $O/maxsim_batch.log; cp gpurun_out/prof_maxsim/summary.json $O/r03_maxsim_batch64.json
cd /tmp
rm -rf $R/gpurun_out/prof_r03_s3
S3_ONLY_FIRST=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03_s3 -- python3 $R/tools/s3_forward_probe.py > $O/s3_forward.json 2> $O/s3_forward.err
cp $R/gpurun_out/prof_r03_s3/*/*kernel_stats.csv $O/r03_s3_forward_kernel_stats.csv 2>/dev/null
tail -1 $O/s3_forward.json
ls $O
