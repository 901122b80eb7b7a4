R=$GRAFT_REPO_ROOT; cd $R; S=$R/gpurun_out/r4_secondary2; mkdir -p $S
timeout 600 python tools/bench_dmd_step.py > $S/dmd_step.txt 2>&1
timeout 600 python tools/bench_diffusion_step.py > $S/diffusion_step.txt 2>&1
timeout 600 python tools/bench_gan_step.py > $S/gan_step.txt 2>&1
timeout 600 python tools/bench_vit_train.py > $S/vit_train.txt 2>&1
timeout 600 python tools/bench_sample.py > $S/sample.txt 2>&1
tail -n 4 $S/*.txt
