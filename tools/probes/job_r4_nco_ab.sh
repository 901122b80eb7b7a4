cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r4_nco_ab; mkdir -p $OUT
for r in 1 2 3; do for v in 0 1; do
  timeout 600 python tools/probes/ab_flag.py dmvae_amd.functional NORM_CONV_OUT_FUSED_BWD $v --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused=$v', d['ms_per_step'], d['ms_per_step_windows'], d['env']['sclk_mhz_avg'])" | tee -a $OUT/ab.txt
done; done
