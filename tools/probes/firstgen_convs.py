"""Shapes of the C2 step that run on the first-generation kernels (conv_fwd.hip / conv_wgrad.hip): time per call, for library A/Bs through DMVAE_LIB."""
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
bf = torch.bfloat16
for (n, h, cin, cout, ks) in [(32, 16, 512, 512, 3), (32, 32, 512, 32, 3), (32, 32, 32, 32, 3), (8, 32, 128, 128, 3), (32, 8, 512, 512, 3)]:
    x = torch.randn(n, h, h, cin, device="cuda").to(bf); w = (torch.randn(cout, ks * ks, cin, device="cuda") * 0.05).to(bf)
    dy = torch.randn(n, h, h, cout, device="cuda").to(bf)
    t = timed(lambda: ops.conv2d_nhwc(x, w, None, ks=ks)); tw = timed(lambda: ops.conv2d_nhwc_wgrad(dy, x, ks))
    fl = 2.0 * n * h * h * cin * cout * ks * ks
    print(f"[{n},{h},{h}] {cin}>{cout} k{ks}: fwd {t:7.1f} us ({fl/t/1e6:6.1f} TF/s)  wgrad {tw:7.1f} us ({fl/tw/1e6:6.1f} TF/s)")
q = torch.randn(32, 1024, 512, device="cuda").to(bf); k = torch.randn(32, 1024, 512, device="cuda").to(bf)
t = timed(lambda: ops.gemm_nt(q, k, out_f32=True)); print(f"batched NT GEMM 32 x [1024 x 1024 x 512]: {t:7.1f} us ({2.0*32*1024*1024*512/t/1e6:6.1f} TF/s)")
