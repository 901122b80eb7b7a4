cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5_attn2
python -m pytest tests/test_gpu_attention_bwd.py tests/test_gpu_vit_train.py tests/test_gpu_vit_pin.py tests/test_gpu_dit.py tests/test_gpu_dit_stack.py tests/test_gpu_modules.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do
python tools/bench_attention.py 2>&1 | grep -v amdgpu
done | tee gpurun_out/r5_attn2/bench4.txt
