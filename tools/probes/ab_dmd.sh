# same-box A/B of the DMD stage bench: this tree with the cond + uncond evaluations batched / unbatched, and round 2's tree (tools/probes/bin/old_tree, built from commit 216b63a)
R=$GRAFT_REPO_ROOT; cd $R
for i in 1 2; do
echo "== new, DMVAE_DMD_BATCH_CFG=1"; DMVAE_DMD_BATCH_CFG=1 python tools/bench_dmd_step.py 2>&1 | grep "ms/step"
echo "== new, DMVAE_DMD_BATCH_CFG=0"; DMVAE_DMD_BATCH_CFG=0 python tools/bench_dmd_step.py 2>&1 | grep "ms/step"
if [ -d tools/probes/bin/old_tree ]; then echo "== round-2 tree"; (cd tools/probes/bin/old_tree && python tools/bench_dmd_step.py 2>&1 | grep "ms/step"); fi
done
