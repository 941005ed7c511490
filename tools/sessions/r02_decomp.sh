#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
b() { timeout -k 10 300 python bench.py --no-cpu-baseline --no-encode-leg --rows 1250000 --steps 200 --warmup 10 "$@" 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']
print('$*', '| ms/step', d['ms_per_step'], '| kernel', r['avg_kernel_ms'], '|', 'ONE' if 'one launch' in c['search_path'] else 'classic')"; }
b --pipeline on --one-launch
b --pipeline on
b --pipeline on --one-launch --submit-stream null
b --pipeline off --force-exchange
b --pipeline off --force-exchange --classic
b --pipeline on --force-exchange --one-launch
b --pipeline on --force-exchange
