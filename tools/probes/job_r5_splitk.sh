R=$GRAFT_REPO_ROOT; cd $R; OUT=$R/gpurun_out/r5_splitk; mkdir -p $OUT
for r in 1 2; do
echo "SPLITK=3 K>=3072:" | tee -a $OUT/ab3.txt; timeout 600 python tools/bench_dmd_step.py 2>&1 | tail -3 | tee -a $OUT/ab3.txt
echo "SPLITK=3 K>=1152:" | tee -a $OUT/ab3.txt; DMVAE_SPLITK_MINK=1152 timeout 600 python tools/bench_dmd_step.py 2>&1 | tail -3 | tee -a $OUT/ab3.txt
done
