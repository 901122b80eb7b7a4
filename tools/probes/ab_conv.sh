# Same-box A/B of library variants on the conv micro-benchmark.  usage: bash tools/probes/ab_conv.sh <which> names...
R=$GRAFT_REPO_ROOT; cd $R; W=$1; shift
for n in "$@"; do
  if [ "$n" = "tree" ]; then unset DMVAE_LIB; else export DMVAE_LIB=$R/tools/probes/bin/lib_$n.so; fi
  echo "=== $n"; REPS=${REPS:-20} python tools/bench_conv.py $W 2>&1 | grep -v Warn
done
