"""GroupNorm backward with and without the swish derivative (register count 110-124 vs 82-110 VGPRs: 4 vs 5-6 waves per SIMD): is the pass limited by occupancy / VALU or by the memory system?"""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from dmvae_amd import ops
def timed(fn, reps=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (n, h, w, c) in [(32, 256, 256, 128), (32, 128, 128, 256), (32, 64, 64, 512)]:
    x = torch.randn(n, h, w, c, device="cuda").to(torch.bfloat16); da = torch.randn_like(x); g, b = torch.randn(c, device="cuda"), torch.randn(c, device="cuda")
    S = x.numel() * 2
    st = ops.groupnorm_stats(x)
    for sw in (True, False):
        t = timed(lambda: ops.groupnorm_bwd(da, x, st, g, b, sw))
        ta = timed(lambda: ops.groupnorm_apply(x, st, g, b, sw))
        print(f"[{n},{h},{w},{c}] swish={sw}: bwd (4+4+2 B/elem) {t:7.1f} us {5*S/t/1e6:5.2f} TB/s | apply {ta:7.1f} us {2*S/ta/1e6:5.2f} TB/s")
