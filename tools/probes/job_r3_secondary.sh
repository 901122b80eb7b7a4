# Round-3 secondary measurements (everything except the headline bench): GEMM kernel vs vendor library (warm / cold operands), ViT-L fwd+bwd, DiT forward,
# DMD cycle, diffusion step, KL / MMD, per-shape step table.  Output: gpurun_out/r3_secondary/*.txt
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3_secondary; mkdir -p $OUT; cd $R
python tools/bench_gemm.py --rounds 7 > $OUT/gemm_warm.txt 2>&1
python tools/bench_gemm.py --rounds 7 --cold > $OUT/gemm_cold.txt 2>&1
python tools/bench_vit_train.py > $OUT/vit_train.txt 2>&1
python tools/bench_dit.py > $OUT/dit_fwd.txt 2>&1
python tools/bench_dmd_step.py > $OUT/dmd_step.txt 2>&1
python tools/bench_diffusion_step.py > $OUT/diffusion_step.txt 2>&1
python tools/bench_klmmd.py > $OUT/klmmd.txt 2>&1
python tools/bench_gan_step.py > $OUT/gan_step.txt 2>&1
python tools/step_shapes.py > $OUT/step_shapes.txt 2>&1
tail -n 4 $OUT/*.txt
