#!/bin/bash
# final check of the tree: whole -m gpu suite, smoke, a 4-rank gloo rehearsal of bench.py through its own launcher, the default bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03w; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/pytest.log; tail -4 $O/pytest.log
timeout -k 10 120 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke.log
TS_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 4 --steps 10 --warmup 2 --rows 4000000 --no-encode-leg > $O/bench_gloo4.json 2> $O/bench_gloo4.err; echo "gloo4 rc=$?"; cut -c1-400 $O/bench_gloo4.json
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], r['frac'], r['measured_read_peak'], r['frac_of_measured_read_peak'], r['traffic']); print({k:v for k,v in d['secondary'].items() if 'qps' in k or 'error' in k}); print(d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
