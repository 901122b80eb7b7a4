R=$GRAFT_REPO_ROOT; cd $R
for n in tree gnnt1 gnnt2 gnnt3; do
  if [ "$n" = "tree" ]; then unset DMVAE_LIB; else export DMVAE_LIB=$R/tools/probes/bin/lib_$n.so; fi
  echo "=== $n"; python tools/bench_gn.py 2>&1 | grep "^\["
done
