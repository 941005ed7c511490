#!/bin/bash
# rocprofv3 runs for profiles/: kernel trace + stats, then PMC passes (separate runs).
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r01e
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/trace_bench.log 2>&1
echo "trace rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1
echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/pmc_write.log 2>&1
echo "write rc=$?"
find $OUT -name "*.csv" | head -20
