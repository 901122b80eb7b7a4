# One-call evidence job of a round: full -m gpu suite, rocprofv3 kernel trace + stats of the bench command, plain bench, smoke, PMC traffic passes of the
# dominant kernels, secondary benches.  usage: bash tools/job_evidence.sh <round number>      -> gpurun_out/r<N>_evidence/
R=$GRAFT_REPO_ROOT; N=${1:-4}; cd $R; E=$R/gpurun_out/r${N}_evidence; mkdir -p $E
bash tools/gpu_job.sh r${N}_evidence/job > $E/job.log 2>&1
bash tools/pmc_traffic_r3.sh r${N}_evidence/traffic $N > $E/traffic.log 2>&1
S=$E/secondary; mkdir -p $S
timeout 600 python tools/bench_gemm.py --rounds 7 > $S/gemm_warm.txt 2>&1
timeout 600 python tools/bench_gemm.py --rounds 7 --cold > $S/gemm_cold.txt 2>&1
timeout 600 python tools/bench_gemm.py --rounds 7 --cold --kmajor > $S/gemm_cold_kmajor.txt 2>&1
timeout 600 python tools/bench_vit_train.py > $S/vit_train.txt 2>&1
timeout 600 python tools/bench_dit.py > $S/dit_fwd.txt 2>&1
timeout 600 python tools/bench_dmd_step.py > $S/dmd_step.txt 2>&1
timeout 600 python tools/bench_diffusion_step.py > $S/diffusion_step.txt 2>&1
timeout 600 python tools/bench_klmmd.py > $S/klmmd.txt 2>&1
timeout 600 python tools/bench_gan_step.py > $S/gan_step.txt 2>&1
timeout 600 python tools/bench_gn.py > $S/gn.txt 2>&1
timeout 600 python tools/step_shapes.py > $S/step_shapes.txt 2>&1
timeout 600 python tools/bench_sample.py > $S/sample.txt 2>&1
tail -25 $E/job.log; tail -5 $E/traffic.log; tail -n 3 $S/*.txt
