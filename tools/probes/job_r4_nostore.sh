# timing experiment (wrong results on purpose): the direct epilogue of the halo conv instantiations without its stores, per kernel on random operands (tools/bench_conv.py) --
# the upper bound of what hiding the write burst could buy.  (Inside the step the variant is confounded: without stores the activations are stale memory, the clock rises to 2.32 GHz at 1.07 kW.)
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r4_nostore; mkdir -p $OUT; export TMPDIR=/tmp
for r in 1 2; do
  for v in base nostore; do
    if [ $v = nostore ]; then export DMVAE_LIB=$GRAFT_REPO_ROOT/tools/probes/bin/lib_nostore.so; else unset DMVAE_LIB; fi
    echo "== $v" | tee -a $OUT/conv.txt
    REPS=30 timeout 600 python tools/bench_conv.py fwd,res 2>/dev/null | tee -a $OUT/conv.txt
  done
done
