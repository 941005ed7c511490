#!/bin/bash
# Round-2 final session, part A: the full GPU suite, the headline bench (with cpu_baseline and the secondary leg),
# rocprofv3 kernel trace + PMC passes of the same command, one-launch timeline.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02f
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -q --timeout 600 > $O/gputest.log 2>&1
tail -2 $O/gputest.log
timeout -k 10 400 python bench.py > $O/bench.json 2>$O/bench.err
tail -c 600 $O/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-encode-leg > $O/trace_bench.log 2>&1
echo "trace rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-encode-leg > $O/pmc_fetch.log 2>&1
echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-encode-leg > $O/pmc_write.log 2>&1
echo "write rc=$?"
# the same three passes for the one-launch kernel at the 8-GPU shard shape (its default regime)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_1p25 -- python3 $R/bench.py --rows 1250000 --steps 100 --warmup 5 --no-cpu-baseline --no-encode-leg > $O/trace_1p25_bench.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_1p25 -- python3 $R/bench.py --rows 1250000 --steps 10 --warmup 1 --no-cpu-baseline --no-encode-leg > $O/pmc_fetch_1p25.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_1p25 -- python3 $R/bench.py --rows 1250000 --steps 10 --warmup 1 --no-cpu-baseline --no-encode-leg > $O/pmc_write_1p25.log 2>&1
echo "1p25 profiles done"
cd $R
bash tools/trace_fused.sh 1250000 > $O/trace_fused.log 2>&1
python3 tools/summarize_r02.py $O > $O/summary.log 2>&1
tail -40 $O/summary.log
