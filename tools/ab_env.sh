# Same-box A/B of environment settings on the bench: bash tools/ab_env.sh "VAR=1" "VAR=0" ...   (each argument: space-separated assignments, or "-" for none)
R=$GRAFT_REPO_ROOT; cd $R
for cfg in "$@"; do
  echo "=== $cfg"
  ( if [ "$cfg" != "-" ]; then export $cfg; fi; python bench.py --steps 15 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'avg_us', d['roofline']['avg_launch_us'], 'all_conv', d['roofline']['all_conv_fwd_dgrad_launches']['achieved_TFLOPs'], 'wgrad', d['roofline_wgrad']['frac'])" )
done
