# Same-box A/B of library builds: for each lib in tools/probes/bin/lib_*.so (and the in-tree one) run the phase probe and a short bench.  usage: bash tools/ab_libs.sh [names...]
R=$GRAFT_REPO_ROOT; cd $R
for n in "$@"; do
  if [ "$n" = "tree" ]; then unset DMVAE_LIB; else export DMVAE_LIB=$R/tools/probes/bin/lib_$n.so; fi
  echo "=== $n"
  true
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'avg_us', d['roofline']['avg_launch_us'], 'wgrad', d['roofline_wgrad']['frac'])"
done
