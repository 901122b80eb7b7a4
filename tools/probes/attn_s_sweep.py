"""ViT attention forward (B = 32, 16 heads of 64): time against the token count -- what does the ninth 32-query block (the class token: S = 257) cost?"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from dmvae_amd import ops
def t(fn, reps=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for s in (128, 224, 256, 257, 288):
    qkvs = [torch.randn(32, s, 3 * 16 * 64, device="cuda").to(torch.bfloat16) for _ in range(12)]     # 12 x 50 MB: cold inputs like in the step
    i = [0]
    def f():
        i[0] = (i[0] + 1) % 12
        ops.attention_qkv(qkvs[i[0]], 16, 0.125)
    print(f"S={s}: fwd {t(f):.1f} us (cold inputs)   {t(lambda: ops.attention_qkv(qkvs[0], 16, 0.125)):.1f} us (warm)")
