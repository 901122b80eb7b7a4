R=$GRAFT_REPO_ROOT; cd $R; OUT=$R/gpurun_out/r5_attn; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_attention_bwd.py tests/test_gpu_dit_stack.py tests/test_gpu_dit.py tests/test_gpu_vit_train.py tests/test_gpu_kernels.py -q --tb=short -x 2>&1 | tail -12
timeout 1200 python -m pytest tests -m gpu -x -q -k "dmd or diffusion or trainer or sampler or step" 2>&1 | tail -3
python - <<'PY' 2>&1 | tee $OUT/attn_bwd_us.txt
import torch, time
from dmvae_amd import ops
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
for b in (16, 64):
    h,n,d,dp=16,256,72,96
    q=torch.zeros(b*h,n,dp,device='cuda',dtype=torch.bfloat16); q[...,:d].normal_(); k=torch.zeros_like(q); k[...,:d].normal_()
    v=torch.randn(b*h,n,d,device='cuda').bfloat16(); do=torch.randn(b,n,h*d,device='cuda').bfloat16()
    o,lse=ops.attention_heads(q,k,v,b,d**-0.5,need_lse=True)
    print(f"DiT heads B={b}: fwd {t(lambda: ops.attention_heads(q,k,v,b,d**-0.5)):.1f} us, fwd+lse {t(lambda: ops.attention_heads(q,k,v,b,d**-0.5,need_lse=True)):.1f}, bwd plain {t(lambda: ops.attention_bwd_heads(q,k,v,o,do,b,d**-0.5)):.1f} us, bwd lse {t(lambda: ops.attention_bwd_heads(q,k,v,o,do,b,d**-0.5,lse=lse)):.1f} us")
b,h,s=16,16,257
qkv=torch.randn(b,s,3*h*64,device='cuda').bfloat16(); do=torch.randn(b,s,h*64,device='cuda').bfloat16()
o,lse=ops.attention_qkv(qkv,h,0.125,need_lse=True)
print(f"ViT qkv B={b}: bwd plain {t(lambda: ops.attention_bwd_qkv(qkv,o,do,h,0.125)):.1f} us, bwd lse {t(lambda: ops.attention_bwd_qkv(qkv,o,do,h,0.125,lse=lse)):.1f} us")
PY
timeout 600 python tools/bench_dmd_step.py 2>&1 | tail -4 | tee $OUT/dmd_step.txt
ONLY=hip timeout 600 python tools/bench_diffusion_step.py 2>&1 | tail -1 | cut -c1-200 | tee $OUT/diffusion_step.txt
