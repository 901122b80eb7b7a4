# Measure hipBLASLt solution tables (PyTorch TunableOp) for the library GEMMs of the stages beside the C2 bench: diffusion step (C4), DMD step (C3), trainable ViT-L.
# usage (GPU box): bash tools/tune_gemms.sh <tag>   -> gpurun_out/<tag>/*.csv ; copy the tables to keep into dmvae_amd/tuned/
R=$GRAFT_REPO_ROOT; TAG=${1:-tune}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
export PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=30 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=5
for t in diffusion_step dmd_step vit_train; do
  PYTORCH_TUNABLEOP_FILENAME=$OUT/$t.csv STEPS=2 timeout 1500 python tools/bench_$t.py > $OUT/$t.log 2>&1
  tail -3 $OUT/$t.log
done
ls -la $OUT
