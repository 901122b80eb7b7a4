# kernel times of the KL-only pass on a 268 MB tensor for library variants: bash tools/probes/klpass_time.sh tree klu8 ...
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
for n in "$@"; do
  if [ "$n" = "tree" ]; then unset DMVAE_LIB; else export DMVAE_LIB=$R/tools/probes/bin/lib_$n.so; fi
  rm -rf /tmp/klp; cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/klp -o t -- python $R/tools/probes/klpass.py > /dev/null 2>&1
  echo "== $n"; python - <<'PY'
import csv, glob
f = glob.glob("/tmp/klp/**/t_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "kl_" in r["Name"]: print("  %-40s %8.1f us x %s" % (r["Name"][:40], float(r["AverageNs"]) / 1e3, r["Calls"]))
PY
done
