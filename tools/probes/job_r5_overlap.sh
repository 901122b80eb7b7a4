R=$GRAFT_REPO_ROOT; cd $R; OUT=$R/gpurun_out/r5_overlap; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -x -q -k "dmd or diffusion or trainer or sampler or step or dit or vit_train" 2>&1 | tail -6
for r in 1 2; do
  echo "plan over 10 tiles (default):" | tee -a $OUT/tiles_ab.txt; timeout 600 python tools/bench_dmd_step.py 2>&1 | tail -3 | tee -a $OUT/tiles_ab.txt
  echo "DMVAE_GEMM_NPLAN=12:" | tee -a $OUT/tiles_ab.txt; DMVAE_GEMM_NPLAN=12 timeout 600 python tools/bench_dmd_step.py 2>&1 | tail -3 | tee -a $OUT/tiles_ab.txt
done
ONLY=hip timeout 600 python tools/bench_diffusion_step.py 2>&1 | tail -1 | cut -c1-200 | tee $OUT/diffusion_step.txt
