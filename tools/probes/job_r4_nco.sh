cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r4_nco; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_norm_conv_out.py -x -q > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
timeout 300 python tools/probes/time_norm_conv_out.py > $OUT/time.log 2>&1; cat $OUT/time.log
export TMPDIR=/tmp; cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o t -- python $GRAFT_REPO_ROOT/tools/probes/time_norm_conv_out.py > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1; cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r4_nco/prof/**/*kernel_stats.csv",recursive=True)
for r in list(csv.DictReader(open(f[0])))[:8]: print(r["Name"][:90], r["Calls"], r["AverageNs"])
PY
find $OUT/prof -name '*kernel_trace.csv' -delete
