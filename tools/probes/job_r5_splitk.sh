R=$GRAFT_REPO_ROOT; cd $R; OUT=$R/gpurun_out/r5_splitk; mkdir -p $OUT
for r in 1 2; do
echo "SPLITK=3 K>=3072:" | tee -a $OUT/ab2.txt; timeout 600 python tools/bench_dmd_step.py 2>&1 | tail -3 | tee -a $OUT/ab2.txt
echo "SPLITK=3 K>=6144:" | tee -a $OUT/ab2.txt; DMVAE_SPLITK_MINK=6144 timeout 600 python tools/bench_dmd_step.py 2>&1 | tail -3 | tee -a $OUT/ab2.txt
echo "SPLITK off:" | tee -a $OUT/ab2.txt; DMVAE_SPLITK=0 timeout 600 python tools/bench_dmd_step.py 2>&1 | tail -3 | tee -a $OUT/ab2.txt
done
