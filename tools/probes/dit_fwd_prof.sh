# rocprofv3 kernel stats of the LightningDiT inference forward (B = 16), this tree and round 2's tree, same box
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; OUT=$R/gpurun_out/dit_fwd_prof; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/new -o t -- python $R/tools/bench_dit.py > $OUT/new.log 2>&1
if [ -d $R/tools/probes/bin/old_tree ]; then rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/old -o t -- python $R/tools/probes/bin/old_tree/tools/bench_dit.py > $OUT/old.log 2>&1; fi
rm -f $OUT/*/t_kernel_trace.csv
for t in new old; do echo "== $t"; grep "forward" $OUT/$t.log; python - $OUT/$t/t_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:22]:
    print(f'{float(r["TotalDurationNs"])/1e6:9.2f} ms {int(r["Calls"]):6d} calls {float(r["AverageNs"])/1e3:8.1f} us  {r["Name"][:110]}')
PY
done
