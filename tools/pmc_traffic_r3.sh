# HBM traffic of the three MFMA kernels that carry the step -- the dominant conv_pp instantiation and both wgrad_pp instantiations -- inside the real step:
# one counter per --pmc pass (kernel-trace only), dispatches filtered to those kernels' names so that the other ~700 kernels of the step run unprofiled.
# usage: bash tools/pmc_traffic_r3.sh <tag> [round]      -> gpurun_out/<tag>/{conv_pp_traffic.json, wgrad_pp_traffic.json, traffic_summary.txt}
R=$GRAFT_REPO_ROOT; TAG=$1; ROUND=${2:-3}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
RX='conv_pp_kernel<256, 256, 2, 4, 4, false, false, true, false, false, false, false, true>|wgrad_pp_kernel<2, 2, 2, 4, false, false, false, 32>|wgrad_pp_kernel<1, 3, 2, 4, false, true, false, 64>'
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $C --kernel-trace --kernel-include-regex "$RX" --output-format csv -d $OUT/$C -o t -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/$C.log 2>&1
  echo "$C rc=$?"
done
cd $R && python tools/pmc_traffic_json.py $OUT $ROUND | tee $OUT/traffic_summary.txt
rm -rf $OUT/FETCH_SIZE/*kernel_trace* $OUT/WRITE_SIZE/*kernel_trace*
