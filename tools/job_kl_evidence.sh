# KL / MMD evidence: rocprofv3 kernel stats of tools/bench_kl_shapes.py, then FETCH_SIZE / WRITE_SIZE in separate --pmc passes (kernel-trace only).
# usage: bash tools/job_kl_evidence.sh <tag>   -> gpurun_out/<tag>/{bench.txt, stats.csv, pmc_summary.txt}
R=$GRAFT_REPO_ROOT; TAG=${1:-kl}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
python $R/tools/bench_kl_shapes.py > $OUT/bench.txt 2>&1; cat $OUT/bench.txt
: > $OUT/pmc_summary.txt
for S in kl268 g32 g1024; do
  SHAPES=$S REPS=10 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$S -o kl -- python $R/tools/bench_kl_shapes.py > $OUT/stats_$S.log 2>&1
  grep -E '"Name"|dmvae_loss' $(ls $OUT/stats_$S/*/kl_kernel_stats.csv $OUT/stats_$S/kl_kernel_stats.csv 2>/dev/null | head -1) > $OUT/stats_$S.csv
  for C in FETCH_SIZE WRITE_SIZE; do
    SHAPES=$S REPS=2 timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/${C}_$S -o t -- python $R/tools/bench_kl_shapes.py > $OUT/${C}_$S.log 2>&1
  done
  echo "== $S (tools/bench_kl_shapes.py SHAPES=$S; separate --pmc passes, kernel-trace only)" >> $OUT/pmc_summary.txt
  (cd $R && python tools/pmc_kl_summary.py $(ls $OUT/FETCH_SIZE_$S/*/t_counter_collection.csv $OUT/FETCH_SIZE_$S/t_counter_collection.csv 2>/dev/null | head -1) $(ls $OUT/WRITE_SIZE_$S/*/t_counter_collection.csv $OUT/WRITE_SIZE_$S/t_counter_collection.csv 2>/dev/null | head -1)) >> $OUT/pmc_summary.txt 2>&1
  rm -rf $OUT/stats_$S $OUT/FETCH_SIZE_$S $OUT/WRITE_SIZE_$S
done
cat $OUT/pmc_summary.txt; head -5 $OUT/stats_kl268.csv | cut -c1-160
