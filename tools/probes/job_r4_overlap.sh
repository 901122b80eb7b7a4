# Round 4: weight gradients on a side stream next to the backward chain -- same-box A/B on the bench.  usage: bash tools/probes/job_r4_overlap.sh
R=$GRAFT_REPO_ROOT; cd $R; OUT=$R/gpurun_out/r4_overlap; mkdir -p $OUT
( DMVAE_WGRAD_STREAM=1 DMVAE_WGRAD_CUS=192 DMVAE_PP_DYNAMIC=1 timeout 900 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -5 ) > $OUT/pytest_side.log 2>&1
tail -3 $OUT/pytest_side.log
bash tools/ab_env.sh "-" \
  "DMVAE_WGRAD_STREAM=1" \
  "DMVAE_WGRAD_STREAM=1 DMVAE_PP_DYNAMIC=1" \
  "DMVAE_WGRAD_STREAM=1 DMVAE_WGRAD_CUS=224 DMVAE_PP_DYNAMIC=1" \
  "DMVAE_WGRAD_STREAM=1 DMVAE_WGRAD_CUS=192 DMVAE_PP_DYNAMIC=1" \
  "DMVAE_WGRAD_STREAM=1 DMVAE_WGRAD_CUS=192" \
  "DMVAE_WGRAD_STREAM=1 DMVAE_WGRAD_CUS=160 DMVAE_PP_DYNAMIC=1" \
  "DMVAE_WGRAD_STREAM=1 DMVAE_WGRAD_CUS=128 DMVAE_PP_DYNAMIC=1" \
  "DMVAE_WGRAD_STREAM=1 DMVAE_WGRAD_CUS=192 DMVAE_PP_DYNAMIC=1 DMVAE_WGRAD_STREAM_PRIO=-1" \
  "DMVAE_PP_DYNAMIC=1" \
  "-" 2>&1 | tee $OUT/ab.log
