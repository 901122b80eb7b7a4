cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6_soak
STEPS=200 timeout 900 python tools/probes/soak_dmd.py 2>&1 | grep -v Warning | tail -12 | tee gpurun_out/r6_soak/soak_dmd.txt
STEPS=300 DISC_START=100 B=32 timeout 900 python tools/soak.py 2>&1 | grep -v Warning | tail -14 | tee gpurun_out/r6_soak/soak_tokenizer_gan.txt
