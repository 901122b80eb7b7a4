R=$GRAFT_REPO_ROOT; cd $R; OUT=$R/gpurun_out/r5_wgab; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_dit_stack.py -q -x -k "grouped" 2>&1 | tail -2
for r in 1 2; do
echo "grouped problems aligned to 8 blocks (tree):" | tee -a $OUT/ab.txt; timeout 600 python tools/bench_dmd_step.py 2>&1 | tail -3 | tee -a $OUT/ab.txt
echo "unaligned (previous build):" | tee -a $OUT/ab.txt; DMVAE_LIB=$R/tools/probes/bin/lib_oldwg.so timeout 600 python tools/bench_dmd_step.py 2>&1 | tail -3 | tee -a $OUT/ab.txt
done
echo "C4 tree:" | tee -a $OUT/ab.txt; ONLY=hip timeout 600 python tools/bench_diffusion_step.py 2>&1 | tail -1 | cut -c1-100 | tee -a $OUT/ab.txt
echo "C4 previous build:" | tee -a $OUT/ab.txt; DMVAE_LIB=$R/tools/probes/bin/lib_oldwg.so ONLY=hip timeout 600 python tools/bench_diffusion_step.py 2>&1 | tail -1 | cut -c1-100 | tee -a $OUT/ab.txt
bash tools/pmc_stage_traffic.sh r5_wgab dmd 'wgrad_pp_grouped_kernel' 2>&1 | tail -4
