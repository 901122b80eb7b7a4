R=$GRAFT_REPO_ROOT; cd $R; OUT=$R/gpurun_out/r5_gan; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_modules.py -q --tb=short -k "shortcut_in_norm or tail_stage" 2>&1 | tail -8
cd /tmp
STAGE=gan CYCLES=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_gan -o gan -- python $R/tools/prof_stage.py > $OUT/prof_gan.log 2>&1
T=$(ls $OUT/prof_gan/*/gan_kernel_trace.csv $OUT/prof_gan/gan_kernel_trace.csv 2>/dev/null | head -1)
python $R/tools/stage_trace_summary.py $T 70 > $OUT/gan_trace_summary.txt 2>&1
rm -f $T
head -90 $OUT/gan_trace_summary.txt | cut -c1-200
