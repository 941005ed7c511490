#!/bin/bash
# round 3, first GPU session: the whole -m gpu suite, then the default bench line (live PMC traffic, read probe, pipeline legs)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03s; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
timeout -k 10 120 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke.log
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"
cat $O/bench.json
tail -5 $O/bench.err
