#!/bin/bash
# every number DESIGN.md 4.7 / 4.5 quote for the written-out forwards, in one session
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out/r02e; mkdir -p $O
timeout -k 10 200 python tools/attn_probe.py > $O/attention_varlen.jsonl 2>>$O/err.log; echo "attn rc=$?"
timeout -k 10 500 python tools/encoder_rate.py > $O/encoder_rate.jsonl 2>>$O/err.log; echo "rate rc=$?"
bash tools/sessions/r02_s3_profile.sh > $O/s3_profile.txt 2>&1; echo "s3 rc=$?"
bash tools/sessions/r02_pipeline_lines.sh > $O/pipeline_lines.txt 2>&1; echo "lines rc=$?"
cp gpurun_out/r02c/pipeline.jsonl $O/pipeline.jsonl
cp gpurun_out/r02_s3_prof/top.txt $O/s3_forward_kernels.txt; cp gpurun_out/r02_s3_prof/probe.json $O/s3_forward.json
cp $(find gpurun_out/r02_s3_prof/trace -name "*_kernel_stats.csv" | head -1) $O/s3_forward_kernel_stats.csv
tail -12 $O/pipeline_lines.txt
