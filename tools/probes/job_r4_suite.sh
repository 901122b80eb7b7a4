R=$GRAFT_REPO_ROOT; cd $R; OUT=$R/gpurun_out/r4_suite; mkdir -p $OUT; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -v "^E *frame\|^frame\|Warning\|warnings.warn\|^  *lp = LPIPS\|^tests/" | tail -40 > $OUT/pytest_gpu.log; tail -6 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
