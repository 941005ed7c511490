#!/bin/bash
# Round-2 measurement session (one gpurun call): per-rank shard shapes with the exchange (one-launch vs
# five-launch filter path), the headline bench, MaxSim, the full pipeline, encoder rates.
# Every line goes to gpurun_out/r02/*.log|jsonl; the summaries worth keeping are copied to profiles/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02
mkdir -p $O
cd $R
b() { timeout -k 10 300 python bench.py --no-cpu-baseline --no-encode-leg "$@" 2>>$O/err.log | tail -1; }
echo "== shapes with the exchange (RCCL world of one, pipelined)" > $O/progress.log
: > $O/shapes_exchange.jsonl
for rows in 1250000 2500000 5000000; do
  steps=$((200000000 / rows)); [ $steps -gt 200 ] && steps=200
  b --rows $rows --steps $steps --warmup 10 --force-exchange --pipeline on >> $O/shapes_exchange.jsonl
  echo "exchange $rows done" >> $O/progress.log
done
TS_BENCH_CLASSIC=1 b --rows 1250000 --steps 200 --warmup 10 --force-exchange --pipeline on > $O/shapes_exchange_classic.jsonl
echo "== async, no exchange" >> $O/progress.log
: > $O/shapes.jsonl
b --rows 1250000 --steps 200 --warmup 10 >> $O/shapes.jsonl
b --rows 1250000 --steps 200 --warmup 10 --pipeline on >> $O/shapes.jsonl
TS_BENCH_CLASSIC=1 b --rows 1250000 --steps 200 --warmup 10 >> $O/shapes.jsonl
b --rows 1250000 --steps 100 --warmup 10 --sync >> $O/shapes.jsonl
TS_BENCH_CLASSIC=1 b --rows 1250000 --steps 100 --warmup 10 --sync >> $O/shapes.jsonl
b --rows 6250000 --dim 1024 --dtype bf16 --steps 40 --warmup 5 >> $O/shapes.jsonl
b --rows 10000000 --dtype f32 --steps 10 --warmup 2 >> $O/shapes.jsonl
echo "shapes done" >> $O/progress.log
echo "== headline" >> $O/progress.log
timeout -k 10 400 python bench.py > $O/bench.json 2>>$O/err.log
TS_BENCH_CLASSIC=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-encode-leg > $O/bench_classic.json 2>>$O/err.log
echo "headline done" >> $O/progress.log
echo "== maxsim" >> $O/progress.log
timeout -k 10 200 python tools/bench_maxsim.py > $O/maxsim_single.json 2>>$O/err.log
timeout -k 10 200 python tools/bench_maxsim.py --batch 64 --no-check > $O/maxsim_batch64.json 2>>$O/err.log
timeout -k 10 200 python tools/bench_maxsim.py --dtype f32 > $O/maxsim_single_f32.json 2>>$O/err.log
echo "== pipeline" >> $O/progress.log
timeout -k 10 300 python bench_pipeline.py --queries 256 --store --many 64 > $O/pipeline_many.json 2>>$O/err.log
timeout -k 10 300 python bench_pipeline.py --queries 256 --store --many 64 --ids > $O/pipeline_many_ids.json 2>>$O/err.log
timeout -k 10 300 python bench_pipeline.py --queries 256 --store --many 64 --ids --bm25 > $O/pipeline_many_ids_bm25.json 2>>$O/err.log
timeout -k 10 300 python bench_pipeline.py --queries 64 --store --graphs > $O/pipeline_per_query_graphs.json 2>>$O/err.log
echo "== probes" >> $O/progress.log
timeout -k 10 300 python tools/sdpa_probe.py > $O/sdpa_probe.json 2>>$O/err.log
timeout -k 10 400 python tools/encoder_rate.py > $O/encoder_rate.jsonl 2>>$O/err.log
echo "all done" >> $O/progress.log
tail -c 3000 $O/err.log
