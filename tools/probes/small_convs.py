import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dmvae_amd import ops
def timed(fn, reps=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
bf = torch.bfloat16
def conv(n, h, cin, cout, f32=False, act=0, rows_pad=0):
    x = torch.randn(n, h, h, cin, device="cuda").to(bf)
    w = (torch.randn(max(cout, rows_pad), 9, cin, device="cuda") * 0.05).to(bf)
    t = timed(lambda: ops.conv2d_nhwc(x, w, None, ks=3, act=act, out_f32=f32))
    byt = x.numel() * 2 + n * h * h * w.shape[0] * (4 if f32 else 2)
    print(f"conv N={n} {h}x{h} {cin}->{w.shape[0]} f32={f32}: {t:8.1f} us   HBM-compulsory {byt/1e6:7.1f} MB -> {byt/t/1e6:5.2f} TB/s   {2*n*h*h*cin*9*w.shape[0]/t/1e6:7.1f} TF/s")
conv(64, 256, 32, 64, act=2)          # VGG conv1_1 fwd (3 real channels padded to 32)
conv(32, 256, 64, 4, f32=True)        # VGG conv1_1 dgrad (recon half), f32 out
conv(64, 256, 64, 64, act=2)          # VGG conv1_2 fwd
conv(32, 256, 64, 64)                 # VGG conv1_2 dgrad
conv(32, 256, 128, 4, f32=True)       # decoder conv_out fwd
conv(32, 256, 32, 128)                # decoder conv_out dgrad (dy padded to 32 channels)
dy = torch.randn(32, 256, 256, 32, device="cuda").to(bf); a = torch.randn(32, 256, 256, 128, device="cuda").to(bf)
t = timed(lambda: ops.conv2d_nhwc_wgrad(dy, a, 3))
print(f"conv_out wgrad (dy 32ch pad, a 128ch): {t:8.1f} us")
