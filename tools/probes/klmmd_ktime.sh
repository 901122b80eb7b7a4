# kernel time of the KL+MMD pair / finish kernels at B=32 for a few column splits (rocprofv3 kernel stats)
export TMPDIR=/tmp; cd /tmp
for cs in 1 2 3 4; do
  rm -rf /tmp/kk; DMVAE_KLMMD_CSPLIT=$cs rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kk -o k -- python $GRAFT_REPO_ROOT/tools/probes/klmmd_b32.py > /dev/null 2>&1
  python - <<PY
import csv,glob
f=(glob.glob("/tmp/kk/*/k_kernel_stats.csv")+glob.glob("/tmp/kk/k_kernel_stats.csv"))[0]
print("csplit $cs:", "; ".join(f'{r["Name"].split("(")[0][-34:]} {float(r["AverageNs"])/1e3:.1f} us' for r in list(csv.DictReader(open(f)))[:3]))
PY
done
