R=$GRAFT_REPO_ROOT; cd $R; OUT=$R/gpurun_out/r4_a; mkdir -p $OUT
ls /sys/class/drm/ > $OUT/sysfs.txt 2>&1; for c in /sys/class/drm/card*/device; do echo $c; cat $c/vendor; ls $c/hwmon/* 2>/dev/null | head -40; cat $c/unique_id 2>/dev/null; done >> $OUT/sysfs.txt 2>&1
rocm-smi --showclocks --showpower --showtemp --json > $OUT/rocm_smi.json 2>&1
timeout 900 python -m pytest tests/test_gpu_modules.py tests/test_gpu_vit_train.py tests/test_gpu_dit.py tests/test_gpu_train_step.py -x -q -s 2>&1 | grep -v "^$" | tail -40 > $OUT/pytest.log
tail -25 $OUT/pytest.log
timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > $OUT/bench.json; python -c "
import json;d=json.loads(open('$OUT/bench.json').read());print({k:d[k] for k in ('ms_per_step','ms_per_step_windows','env')})"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
