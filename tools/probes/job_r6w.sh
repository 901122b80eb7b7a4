cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6w
timeout 300 python tools/probes/time_conv_pp.py 2>&1 | grep -v amdgpu.ids | cut -c1-400 | tee gpurun_out/r6w/stamps.txt
