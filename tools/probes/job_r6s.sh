cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6s
for a in "4096 1152 3072 3 0" "4096 1152 3072 0 1" "4096 1152 3072 2 0" "4096 1152 6144 3 0"; do
  timeout 120 python tools/probes/time_gemm_sk.py $a 2>&1 | tail -6 | tee -a gpurun_out/r6s/stamps.txt
done
timeout 120 python tools/probes/time_gemm_pp.py 4096 1152 3072 2>&1 | tail -4 | tee -a gpurun_out/r6s/stamps.txt
