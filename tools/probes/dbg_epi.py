import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from dmvae_amd import ops
DEV = "cuda"; BF = torch.bfloat16
def run(n, h, w_, cin, cout, ks, res, f32):
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn(n, h, w_, cin, generator=g).to(DEV).to(BF)
    w = (torch.randn(cout, cin, ks, ks, generator=g) * 0.05).to(DEV)
    b = torch.randn(cout, generator=g).to(DEV)
    r = torch.randn(n, h, w_, cout, generator=g).to(DEV).to(BF) if res else None
    wp = ops.pack_conv_weight(w)
    y = ops.conv2d_nhwc(x, wp, b, r, ks=ks, out_f32=f32).float()
    ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.to(BF).float(), b, padding=ks // 2).permute(0, 2, 3, 1)
    if res: ref = ref + r.float()
    err = (y - ref).abs().reshape(-1, cout)
    bad = err > 0.05
    rows = bad.any(1).nonzero().flatten()
    cols = bad.any(0).nonzero().flatten()
    print(f"n{n} {h}x{w_} {cin}->{cout} ks{ks} res={res} f32={f32}: bad elements {int(bad.sum())} of {bad.numel()}; bad rows {rows.numel()} (first {rows[:12].tolist()}), rows%32 set {sorted(set((rows % 32).tolist()))[:40]}; bad couts {cols.numel()} first {cols[:12].tolist()}")
for res in (False, True):
    for f32 in (False, True):
        run(1, 128, 128, 64, 128, 3, res, f32)
        run(1, 128, 128, 64, 256, 3, res, f32)
print("---- detail")
def detail(n, h, w_, cin, cout, ks):
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn(n, h, w_, cin, generator=g).to(DEV).to(BF)
    w = (torch.randn(cout, cin, ks, ks, generator=g) * 0.05).to(DEV)
    b = torch.randn(cout, generator=g).to(DEV)
    r = torch.randn(n, h, w_, cout, generator=g).to(DEV).to(BF)
    wp = ops.pack_conv_weight(w)
    y = ops.conv2d_nhwc(x, wp, b, r, ks=ks, out_f32=True).float().reshape(-1, cout)
    conv = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.to(BF).float(), b, padding=ks // 2).permute(0, 2, 3, 1).reshape(-1, cout)
    rr = r.float().reshape(-1, cout)
    ref = conv + rr
    bad = ((y - ref).abs() > 0.05).nonzero()
    for k in range(0, min(len(bad), 400), 40):
        i, c = bad[k].tolist()
        d = y[i, c] - conv[i, c]   # what was added instead of the residual
        # search which residual element equals d
        m = (rr - d).abs() < 1e-3
        hits = m.nonzero()[:4].tolist()
        print(f"row {i} cout {c}: y {y[i,c]:.4f} ref {ref[i,c]:.4f} conv {conv[i,c]:.4f} res {rr[i,c]:.4f} added {d:.4f}; residual elements equal to it: {hits}")
detail(1, 128, 128, 64, 128, 3)
