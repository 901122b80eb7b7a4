"""What the GroupNorm statistics cost in conv_pp's epilogue (STATS instantiation vs plain), per shape, with and without a residual operand."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from dmvae_amd import ops
def timed(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (n, h, w, cin, cout) in [(32, 256, 256, 128, 128), (32, 128, 128, 256, 256), (32, 64, 64, 512, 512), (32, 256, 256, 256, 128)]:
    x = torch.randn(n, h, w, cin, device="cuda").bfloat16(); wt = (torch.randn(cout, 9, cin, device="cuda") * 0.02).bfloat16(); b = torch.randn(cout, device="cuda")
    r = torch.randn(n, h, w, cout, device="cuda").bfloat16()
    fl = 2.0 * n * h * w * cin * cout * 9
    t0 = timed(lambda: ops.conv2d_nhwc(x, wt, b, ks=3)); t1 = timed(lambda: ops.conv2d_nhwc_gnstats(x, wt, b, ks=3))
    t2 = timed(lambda: ops.conv2d_nhwc(x, wt, b, r, ks=3)); t3 = timed(lambda: ops.conv2d_nhwc_gnstats(x, wt, b, r, ks=3))
    print(f"[{n},{h},{w}] {cin}>{cout}: plain {t0:.1f} us ({fl/t0/1e6:.0f} TF/s) | +stats {t1:.1f} ({(t1/t0-1)*100:+.1f} %) | +residual {t2:.1f} ({(t2/t0-1)*100:+.1f} %) | +residual+stats {t3:.1f} ({(t3/t0-1)*100:+.1f} %)")
