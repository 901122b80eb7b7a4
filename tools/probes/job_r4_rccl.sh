R=$GRAFT_REPO_ROOT; cd $R; OUT=$R/gpurun_out/r4_rccl; mkdir -p $OUT
for i in 1 2; do timeout 1200 python -m pytest tests/test_gpu_attention_bwd.py tests/test_gpu_conv_bounds.py tests/test_gpu_conv_c2_shapes.py tests/test_gpu_conv_halo.py tests/test_gpu_conv_pp.py tests/test_gpu_conv_thin.py tests/test_gpu_dist.py -q 2>&1 | grep -v "^E *frame\|^frame" | tail -60 > $OUT/run$i.log; tail -3 $OUT/run$i.log; done
grep -n "HIP error\|hipError\|Memory access\|fault\|terminate\|what()" $OUT/run*.log | head -20
