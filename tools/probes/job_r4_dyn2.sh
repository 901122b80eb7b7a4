R=$GRAFT_REPO_ROOT; cd $R; OUT=$R/gpurun_out/r4_dyn2; mkdir -p $OUT
for i in 1 2 3; do timeout 900 python -m pytest tests/test_gpu_dist.py -x -q 2>&1 | grep -v "^frame\|^E *frame" | tail -25; done > $OUT/log.txt 2>&1
grep -n "passed\|failed\|Error\|error" $OUT/log.txt | head -20
