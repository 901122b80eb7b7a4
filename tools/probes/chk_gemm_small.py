import sys, torch
sys.path.insert(0, "/root/repo")
from dmvae_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
for m in (257, 2056, 8224, 256, 64, 100):
    for n, k in ((3072, 1024), (1024, 4096), (2048, 1024), (32, 2048)):
        x = torch.randn(m, k, device="cuda", generator=g).to(torch.bfloat16); w = (torch.randn(n, k, device="cuda", generator=g) * k ** -0.5).to(torch.bfloat16); b = torch.randn(n, device="cuda", generator=g)
        ref = torch.addmm(b.double(), x.double(), w.double().t())
        wk = ops.pack_conv_weight(w.float(), kmajor=True)._dmvae_kmajor.view(k // 32, n, 32)
        for name, ww in (("row", w), ("kmaj", wk)):
            y = ops.linear_bf16(x, ww, b, out_f32=True)
            e = ((y.double() - ref).abs().max() / ref.abs().max()).item()
            print(m, n, k, name, ops.linear_plan(m, n, k), "%.2e" % e, "BAD" if e > 1e-5 else "")
