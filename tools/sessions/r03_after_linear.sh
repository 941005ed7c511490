#!/bin/bash
# round 3: whole -m gpu suite after the projection kernels were rebuilt (storer waves, persistent tiles, projection + LayerNorm),
# the stage-3 forward alone and under rocprofv3, the pipeline lines
R=${GRAFT_REPO_ROOT:-$(pwd)}
export GRAFT_REPO_ROOT=$R
O=$R/gpurun_out/r03n; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/pytest.log; tail -4 $O/pytest.log
timeout -k 10 100 python tools/linear_probe.py > $O/linear_probe.json 2>/dev/null; tail -1 $O/linear_probe.json
timeout -k 10 100 python tools/proj_ln_probe.py > $O/proj_ln_probe.json 2>/dev/null; tail -1 $O/proj_ln_probe.json
timeout -k 10 300 python bench_pipeline.py --queries 256 --store --many 64 --ids > $O/pipeline_many.json 2> $O/pipeline_many.err; tail -1 $O/pipeline_many.json | cut -c1-300
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_r03_s3
S3_ONLY_FIRST=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03_s3 -- python3 $R/tools/s3_forward_probe.py > $O/s3_forward.json 2> $O/s3_forward.err
cp $R/gpurun_out/prof_r03_s3/*/*kernel_stats.csv $O/r03_s3_forward_kernel_stats.csv 2>/dev/null
tail -1 $O/s3_forward.json
