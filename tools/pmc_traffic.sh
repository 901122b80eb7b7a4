# HBM traffic of the conv kernels from the PMC counters (one counter per --pmc pass, kernel-trace only; MI355X_MICROARCH.md "HBM"):
# FETCH_SIZE / WRITE_SIZE per dispatch over the conv microbench on the decoder's shapes (B=32).  usage: bash tools/pmc_traffic.sh <tag>
R=$GRAFT_REPO_ROOT; TAG=$1; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
export REPS=2
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 240 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -o t -- python $R/tools/bench_conv.py fwd > $OUT/$C.log 2>&1
  echo "$C rc=$?"; ls $OUT/$C 2>/dev/null
done
cd $R && python tools/pmc_traffic_summary.py $OUT/FETCH_SIZE/t_counter_collection.csv $OUT/WRITE_SIZE/t_counter_collection.csv > $OUT/traffic_summary.txt; cat $OUT/traffic_summary.txt
