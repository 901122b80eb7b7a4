# adversarial step: ConvK4Fn.backward without the weight gradients of a frozen discriminator, against the previous commit's functional.py (tools/probes/bin/old_tree), one box, alternating
R=$GRAFT_REPO_ROOT; cd $R; OUT=$R/gpurun_out/r4_gan_ab; mkdir -p $OUT
for r in 1 2 3; do
  echo old | tee -a $OUT/ab.txt; (cd tools/probes/bin/old_tree && timeout 600 python tools/bench_gan_step.py 2>&1 | grep with_disc=True | cut -c1-50) | tee -a $OUT/ab.txt
  echo new | tee -a $OUT/ab.txt; timeout 600 python tools/bench_gan_step.py 2>&1 | grep with_disc=True | cut -c1-50 | tee -a $OUT/ab.txt
done
