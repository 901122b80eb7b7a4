# last tree of round 5: soak of the DMD stage (table builds flat, memory flat, losses finite), then the default bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5_final
STEPS=120 timeout 600 python tools/probes/soak_dmd.py > gpurun_out/r5_final/soak.log 2>&1; echo "soak rc=$?"; tail -8 gpurun_out/r5_final/soak.log
timeout 900 python bench.py > gpurun_out/r5_final/bench.log 2>&1; echo "bench rc=$?"; tail -c 3000 gpurun_out/r5_final/bench.log
