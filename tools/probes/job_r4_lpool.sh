cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r4_lpool; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_lpips.py -x -q 2>&1 | tail -8
for r in 1 2 3; do for v in 0 1; do
  timeout 600 python tools/probes/ab_flag.py dmvae_amd.utils.lpips DIFF_POOL_FUSED $v --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('diffpool=$v', d['ms_per_step'], d['ms_per_step_windows'], d['env']['sclk_mhz_avg'])" | tee -a $OUT/ab.txt
done; done
