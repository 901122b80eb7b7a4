cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6_soak2
STEPS=150 timeout 900 python tools/probes/soak_diffusion.py 2>&1 | grep -v Warning | tail -8 | tee gpurun_out/r6_soak2/soak_diffusion.txt
STEPS=200 timeout 900 python tools/probes/soak_dmd.py 2>&1 | grep -v Warning | tail -12 | tee gpurun_out/r6_soak2/soak_dmd.txt
