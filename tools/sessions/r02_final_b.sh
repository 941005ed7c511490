#!/bin/bash
# Round-2 final session, part B: shard shapes (with and without the exchange, both filter paths), cfg4 shapes,
# MaxSim profiles, the full pipeline, encoder rates.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02g
mkdir -p $O
cd $R
b() { timeout -k 10 300 python bench.py --no-cpu-baseline --no-encode-leg "$@" 2>>$O/err.log | tail -1; }
: > $O/shapes_exchange.jsonl
for rows in 1250000 2500000 5000000; do
  steps=$((200000000 / rows)); [ $steps -gt 200 ] && steps=200
  b --rows $rows --steps $steps --warmup 10 --force-exchange --pipeline on >> $O/shapes_exchange.jsonl
done
b --rows 1250000 --steps 200 --warmup 10 --force-exchange --pipeline on --one-launch >> $O/shapes_exchange.jsonl
echo "exchange shapes done"
: > $O/shapes.jsonl
b --rows 1250000 --steps 200 --warmup 10 >> $O/shapes.jsonl
b --rows 1250000 --steps 200 --warmup 10 --classic >> $O/shapes.jsonl
b --rows 1250000 --steps 100 --warmup 10 --sync >> $O/shapes.jsonl
b --rows 1250000 --steps 100 --warmup 10 --sync --classic >> $O/shapes.jsonl
b --rows 2500000 --steps 80 --warmup 10 >> $O/shapes.jsonl
b --rows 2500000 --steps 80 --warmup 10 --classic >> $O/shapes.jsonl
b --rows 10000000 --steps 20 --warmup 3 --one-launch >> $O/shapes.jsonl
b --rows 6250000 --dim 1024 --dtype bf16 --steps 40 --warmup 5 >> $O/shapes.jsonl
b --rows 10000000 --dtype f32 --steps 10 --warmup 2 >> $O/shapes.jsonl
b --rows 5183 --dim 384 --dtype f32 --k 100 --steps 200 --warmup 10 >> $O/shapes.jsonl
b --rows 10000000 --batch 1 --steps 20 --warmup 3 --sync >> $O/shapes.jsonl
echo "shapes done"
timeout -k 10 400 python bench.py --rows 50000000 --dim 1024 --dtype bf16 --steps 10 --warmup 2 --no-cpu-baseline --no-encode-leg 2>>$O/err.log | tail -1 > $O/bench_cfg4_50Mx1024_bf16_1gpu.json
echo "cfg4 done"
bash tools/profile_maxsim.sh > $O/maxsim_single_profile.log 2>&1; cp gpurun_out/prof_maxsim/summary.json $O/maxsim_single.json 2>/dev/null
bash tools/profile_maxsim.sh --batch 64 --no-check > $O/maxsim_batch_profile.log 2>&1; cp gpurun_out/prof_maxsim/summary.json $O/maxsim_batch64.json 2>/dev/null
timeout -k 10 200 python tools/bench_maxsim.py --dtype f32 > $O/maxsim_single_f32.json 2>>$O/err.log
echo "maxsim done"
: > $O/pipeline.jsonl
for extra in "--many 64" "--many 64 --ids" "--many 64 --ids --bm25" "--many 64 --ids --keep" "--many 64 --ids --no-lean"; do
  timeout -k 10 300 python bench_pipeline.py --queries 256 --store $extra 2>>$O/err.log | tail -1 >> $O/pipeline.jsonl
done
timeout -k 10 300 python bench_pipeline.py --queries 64 --store --graphs 2>>$O/err.log | tail -1 >> $O/pipeline.jsonl
echo "pipeline done"
timeout -k 10 400 python tools/encoder_rate.py > $O/encoder_rate.jsonl 2>>$O/err.log
echo "all done"
