L=gpurun_out/maxsim_diag.log; rm -f $L
B="timeout -k 10 100 python tools/bench_maxsim.py"
for args in "" "--lq 64" "--docs 8" "--docs 300 --len-lo 20 --len-hi 60"; do
for v in g8 r16g8 g4; do
echo "$v $args" >> $L; TRISTAGE_LIB=$PWD/tristage-rag_amd/variants_$v.so $B $args 2>/dev/null | tail -1 >> $L
done
done
cat $L
