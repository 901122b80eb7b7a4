# Same-box A/B of library variants on the Linear GEMM shapes: bash tools/probes/ab_gemm.sh "<shape filter>" tree gexp1 gexp2 ...   (variants built by build_variant.sh)
R=$GRAFT_REPO_ROOT; cd $R; F=$1; shift
for n in "$@"; do
  if [ "$n" = "tree" ]; then unset DMVAE_LIB; else export DMVAE_LIB=$R/tools/probes/bin/lib_$n.so; fi
  echo "=== $n"
  python tools/bench_gemm.py --shapes "$F" --rounds 5 ${SWEEP:+--sweep} 2>/dev/null | cut -c1-${CUT:-400}
done
