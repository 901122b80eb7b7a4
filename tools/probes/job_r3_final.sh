# Round-3 evidence in one call: full -m gpu suite, rocprofv3 kernel trace + stats of the bench command, plain bench, smoke, PMC traffic passes, secondary benches.
R=$GRAFT_REPO_ROOT; cd $R
bash tools/gpu_job.sh r3i > gpurun_out/r3i_job.log 2>&1
bash tools/pmc_traffic_r3.sh r3i_traffic > gpurun_out/r3i_traffic.log 2>&1
bash tools/probes/job_r3_secondary.sh > gpurun_out/r3_secondary.log 2>&1
python tools/bench_sample.py > gpurun_out/r3_secondary/sample.txt 2>&1
tail -25 gpurun_out/r3i_job.log; tail -5 gpurun_out/r3i_traffic.log; tail -3 gpurun_out/r3_secondary/sample.txt
