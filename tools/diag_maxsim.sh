L=gpurun_out/maxsim_diag.log; rm -f $L
B="timeout -k 10 100 python tools/bench_maxsim.py --batch 64 --no-check"
for g in 32 63 96 125 190 250; do
echo "t256r16g8 batch grid=$g" >> $L; TS_M16_BATCH_GRID=$g TRISTAGE_LIB=$PWD/tristage-rag_amd/variants_t256r16g8.so $B 2>/dev/null | tail -1 >> $L
done
for m in 1 2; do
echo "t256r16g8 single gridmul=$m" >> $L; TS_M16_GRIDMUL=$m TRISTAGE_LIB=$PWD/tristage-rag_amd/variants_t256r16g8.so timeout -k 10 100 python tools/bench_maxsim.py 2>/dev/null | tail -1 >> $L
done
cat $L
