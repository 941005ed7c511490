L=gpurun_out/maxsim_diag.log; rm -f $L
B="timeout -k 10 100 python tools/bench_maxsim.py --batch 64"
for g in 16 32 48 64 96 128 256; do
echo "batch grid=$g" >> $L; TS_M16_BATCH_GRID=$g TRISTAGE_LIB=$PWD/tristage-rag_amd/variants_tune.so $B 2>/dev/null | tail -1 >> $L
done
cat $L
