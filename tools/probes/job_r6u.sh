R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r6u; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for T in 1 0; do
  for C in FETCH_SIZE WRITE_SIZE; do
    DMVAE_ATTN_BWD_TIGHT=$T timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/${C}_$T -o t -- python $R/tools/bench_attention.py > $OUT/${C}_$T.log 2>&1
  done
  echo "== DMVAE_ATTN_BWD_TIGHT=$T (tools/bench_attention.py; separate --pmc passes, kernel-trace only)" >> $OUT/pmc_summary.txt
  (cd $R && python tools/probes/pmc_attention.py $(ls $OUT/FETCH_SIZE_$T/*/t_counter_collection.csv $OUT/FETCH_SIZE_$T/t_counter_collection.csv 2>/dev/null | head -1) $(ls $OUT/WRITE_SIZE_$T/*/t_counter_collection.csv $OUT/WRITE_SIZE_$T/t_counter_collection.csv 2>/dev/null | head -1)) >> $OUT/pmc_summary.txt 2>&1
  rm -rf $OUT/FETCH_SIZE_$T $OUT/WRITE_SIZE_$T
done
cat $OUT/pmc_summary.txt
