cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6_flaky
for i in 1 2 3; do
  timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 | tee -a gpurun_out/r6_flaky/suite_x3.txt
done
