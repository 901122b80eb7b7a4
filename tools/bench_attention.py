"""Attention kernels at the shapes of the DMD / diffusion stages: LightningDiT-XL/1 heads (16 x 72 channels, 256 tokens, q / k padded to 96) at B = 16 / 64 and
ViT-L/16's packed qkv (16 x 64, 257 tokens) at B = 16 / 32: forward (+ row statistics) and the backward on those statistics, us per call and the algorithmic
bytes (every operand once, every result once) over that time.  DMVAE_ATTN_XCD=0 runs the plain block order for an A/B."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dmvae_amd import ops

BF = torch.bfloat16


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


g = torch.Generator(device="cuda").manual_seed(0)
print(f"DMVAE_ATTN_XCD={os.environ.get('DMVAE_ATTN_XCD', '1 (default)')}")
for B in (16, 32, 64):
    H, N, D, DP = 16, 256, 72, 96
    q = torch.zeros(B * H, N, DP, device="cuda", dtype=BF); k = torch.zeros_like(q)
    q[..., :D] = torch.randn(B * H, N, D, device="cuda", generator=g).to(BF); k[..., :D] = torch.randn(B * H, N, D, device="cuda", generator=g).to(BF)
    v = torch.randn(B * H, N, D, device="cuda", generator=g).to(BF)
    scale = D ** -0.5
    out, lse = ops.attention_heads(q, k, v, B, scale, need_lse=True)
    dout = torch.randn(out.shape, device="cuda", generator=g).to(BF)
    tf = timed(lambda: ops.attention_heads(q, k, v, B, scale, need_lse=True))
    tb = timed(lambda: ops.attention_bwd_heads(q, k, v, out, dout, B, scale, lse=lse))
    fb = (2 * q.numel() + v.numel() + out.numel()) * 2 + lse.numel() * 4
    bb = (4 * q.numel() + 2 * v.numel() + 2 * out.numel()) * 2 + lse.numel() * 4
    fl = 4 * B * H * N * N * D
    print(f"DiT heads B={B:3d}: fwd+lse {tf:6.1f} us ({fb / tf * 1e-6:5.2f} TB/s, {fl / tf * 1e-6:5.0f} TF/s)   bwd lse {tb:6.1f} us ({bb / tb * 1e-6:5.2f} TB/s, {2.5 * fl / tb * 1e-6:5.0f} TF/s)")
for B in (16, 32):
    H, N, D = 16, 257, 64
    qkv = torch.randn(B, N, 3 * H * D, device="cuda", generator=g).to(BF)
    scale = D ** -0.5
    out, lse = ops.attention_qkv(qkv, H, scale, need_lse=True)
    dout = torch.randn(out.shape, device="cuda", generator=g).to(BF)
    tf = timed(lambda: ops.attention_qkv(qkv, H, scale, need_lse=True))
    tb = timed(lambda: ops.attention_bwd_qkv(qkv, out, dout, H, scale, lse=lse))
    fb = (qkv.numel() + out.numel()) * 2 + lse.numel() * 4
    bb = (2 * qkv.numel() + 2 * out.numel()) * 2 + lse.numel() * 4
    fl = 4 * B * H * N * N * D
    print(f"ViT qkv   B={B:3d}: fwd+lse {tf:6.1f} us ({fb / tf * 1e-6:5.2f} TB/s, {fl / tf * 1e-6:5.0f} TF/s)   bwd lse {tb:6.1f} us ({bb / tb * 1e-6:5.2f} TB/s, {2.5 * fl / tb * 1e-6:5.0f} TF/s)")
