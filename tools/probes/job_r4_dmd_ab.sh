# DMD / latent-diffusion stage: the gradient-buffer / operand-cast clean-up against the tree before it (tools/probes/bin/old_tree = the three host files at the previous commit, same library)
R=$GRAFT_REPO_ROOT; cd $R; OUT=$R/gpurun_out/r4_dmd_ab; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_dit.py tests/test_gpu_vit_train.py tests/test_gpu_linear_rows.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -4
timeout 900 python -m pytest tests -m gpu -x -q -k "dmd or diffusion or trainer or sampler" 2>&1 | tail -3
for r in 1 2; do
  echo "old:" | tee -a $OUT/ab.txt; (cd tools/probes/bin/old_tree && timeout 600 python tools/bench_dmd_step.py 2>&1 | tail -3) | tee -a $OUT/ab.txt
  echo "new:" | tee -a $OUT/ab.txt; timeout 600 python tools/bench_dmd_step.py 2>&1 | tail -3 | tee -a $OUT/ab.txt
done
echo "old:" | tee -a $OUT/ab.txt; (cd tools/probes/bin/old_tree && timeout 600 python tools/bench_diffusion_step.py 2>&1 | tail -1 | cut -c1-100) | tee -a $OUT/ab.txt
echo "new:" | tee -a $OUT/ab.txt; timeout 600 python tools/bench_diffusion_step.py 2>&1 | tail -1 | cut -c1-100 | tee -a $OUT/ab.txt
