import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from dmvae_amd import ops
def t(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (b, s, h, tag) in [(32, 257, 16, "ViT-L B=32"), (64, 256, 16, "hd64 B=64")]:
    qkv = torch.randn(b, s, 3 * h * 64, device="cuda").to(torch.bfloat16)
    do = torch.randn(b, s, h * 64, device="cuda").to(torch.bfloat16)
    o = ops.attention_qkv(qkv, h, 0.125)
    print(f"{tag}: fwd {t(lambda: ops.attention_qkv(qkv, h, 0.125)):.1f} us  bwd {t(lambda: ops.attention_bwd_qkv(qkv, o, do, h, 0.125)):.1f} us")
b, n, h, d = 64, 256, 16, 72
q = torch.zeros(b * h, n, 96, device="cuda", dtype=torch.bfloat16); q[..., :72] = torch.randn(b * h, n, 72, device="cuda").to(torch.bfloat16)
k = q.clone(); v = torch.randn(b * h, n, 72, device="cuda").to(torch.bfloat16)
do = torch.randn(b, n, h * d, device="cuda").to(torch.bfloat16)
o = ops.attention_heads(q, k, v, b, d ** -0.5)
print(f"DiT-XL B=64 hd72: fwd {t(lambda: ops.attention_heads(q, k, v, b, d ** -0.5)):.1f} us  bwd {t(lambda: ops.attention_bwd_heads(q, k, v, o, do, b, d ** -0.5)):.1f} us")
