R=$GRAFT_REPO_ROOT; cd $R
for i in 1 2; do
echo "== new"; python tools/bench_dmd_step.py 2>&1 | grep "ms/step"
echo "== new, DMVAE_PACK_BATCHED=0"; DMVAE_PACK_BATCHED=0 python tools/bench_dmd_step.py 2>&1 | grep "ms/step"
done
echo "== round-2 tree"; (cd tools/probes/bin/old_tree && python tools/bench_dmd_step.py 2>&1 | grep "ms/step")
