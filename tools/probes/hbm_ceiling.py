"""What plain streaming kernels reach on this part at the GroupNorm kernels' traffic shapes (bf16, [32, 256, 256, 128] = 537 MB per tensor): ATen's
elementwise add (2 reads + 1 write), copy (1 + 1), sum (1 read) -- the ceiling the GroupNorm passes (2+2, 2+2 reads, 2+2(+2)+2) are judged against."""
import torch
def timed(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for shape in [(32, 256, 256, 128), (32, 128, 128, 256), (32, 64, 64, 512)]:
    x = torch.randn(shape, device="cuda").bfloat16(); y = torch.randn(shape, device="cuda").bfloat16(); z = torch.empty_like(x); w = torch.randn(shape, device="cuda").bfloat16()
    S = x.numel() * 2
    t_add = timed(lambda: torch.add(x, y, out=z))
    t_cp = timed(lambda: z.copy_(x))
    t_sum = timed(lambda: x.sum(dtype=torch.float32))
    t_add3 = timed(lambda: torch.addcmul(x, y, w, out=z))
    print(f"{shape} S={S/1e6:.0f} MB: add {t_add:.1f} us {3*S/t_add/1e6:.2f} TB/s | copy {t_cp:.1f} us {2*S/t_cp/1e6:.2f} TB/s | sum {t_sum:.1f} us {S/t_sum/1e6:.2f} TB/s | addcmul(3r+1w) {t_add3:.1f} us {4*S/t_add3/1e6:.2f} TB/s")
