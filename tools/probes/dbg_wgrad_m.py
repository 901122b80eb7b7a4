import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dmvae_amd import ops
for m, k, co in [(98, 4096, 512), (128, 4096, 512), (98, 256, 128), (72, 8192, 32), (98, 4096, 64), (200, 1024, 512), (98, 128, 512)]:
    g = torch.Generator().manual_seed(1)
    dy = torch.randn(m, co, generator=g).cuda().bfloat16()
    a = torch.randn(m, k, generator=g).cuda().bfloat16()
    dw, db = ops.conv2d_nhwc_wgrad(dy.view(1, 1, m, co), a.view(1, 1, m, k), 1)
    ref = dy.float().t() @ a.float()
    err = ((dw.view(co, k) - ref).abs().max() / ref.abs().max()).item()
    eb = ((db - dy.float().sum(0)).abs().max() / dy.float().sum(0).abs().max()).item()
    print(m, k, co, "dw err", err, "db err", eb)
