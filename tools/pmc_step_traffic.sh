# HBM traffic of the dominant kernel inside the real step: one counter per --pmc pass (kernel-trace only), dispatches filtered to the
# kernel's name so that the other ~800 kernels of the step run unprofiled.  usage: bash tools/pmc_step_traffic.sh <tag>
R=$GRAFT_REPO_ROOT; TAG=$1; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --kernel-include-regex "conv_pp_kernel<256, 256, 2, 4, 4, false, false, true, false, false, false, false, true>" --output-format csv -d $OUT/$C -o t -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/$C.log 2>&1
  echo "$C rc=$?"; ls $OUT/$C 2>/dev/null
done
cd $R && python tools/pmc_traffic_summary.py $OUT/FETCH_SIZE/t_counter_collection.csv $OUT/WRITE_SIZE/t_counter_collection.csv | tee $OUT/traffic_summary.txt
