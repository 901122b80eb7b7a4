R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5_icache; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 -L 2>/dev/null | grep -i "icache\|IFETCH" | head -20 > $OUT/avail.txt
timeout 600 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU --kernel-trace --kernel-include-regex 'conv_pp_kernel|wgrad_pp_kernel|gemm_pp_kernel' --output-format csv -d $OUT/p -o t -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/log.txt 2>&1
echo rc=$?
cd $R; python - <<'PY' | tee $OUT/summary.txt
import csv, glob, collections
agg = collections.OrderedDict()
for f in glob.glob("gpurun_out/r5_icache/p/**/t_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-100:]
        d = agg.setdefault(k, collections.Counter()); d[r["Counter_Name"]] += float(r["Counter_Value"]); d["_n_" + r["Counter_Name"]] += 1
for k, d in agg.items():
    n = d["_n_SQ_WAVE_CYCLES"] or 1
    req, hit, miss = d["SQC_ICACHE_REQ"], d["SQC_ICACHE_HITS"], d["SQC_ICACHE_MISSES"]
    print(f"{k}\n   launches {n:4.0f}  icache req {req/n:12.0f} hits {hit/n:12.0f} misses {miss/n:10.0f} ({100*miss/max(req,1):.2f} %)  ifetch {d['SQ_IFETCH']/n:12.0f}  wave_cycles {d['SQ_WAVE_CYCLES']/n:14.0f} wait_inst {d['SQ_WAIT_INST_ANY']/n:14.0f} busy {d['SQ_BUSY_CYCLES']/n:12.0f}")
PY
cat $OUT/avail.txt | head; find $OUT -name "*kernel_trace*" -delete
