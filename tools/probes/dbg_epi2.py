import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from dmvae_amd import ops
DEV = "cuda"; BF = torch.bfloat16
n, h, w_, cin, cout, ks = 1, 128, 128, 64, 128, 3
g = torch.Generator(device="cpu").manual_seed(1)
x = torch.randn(n, h, w_, cin, generator=g).to(DEV).to(BF)
for mode in ("res_only", "conv_only"):
    w = torch.zeros(cout, cin, ks, ks, device=DEV) if mode == "res_only" else (torch.randn(cout, cin, ks, ks, generator=g) * 0.05).to(DEV)
    b = torch.zeros(cout, device=DEV)
    # residual encodes its own position: row + cout / 1000 is not bf16-exact; use two tensors instead
    if mode == "res_only":
        rows = torch.arange(n * h * w_, device=DEV).view(-1, 1).expand(-1, cout)
        cols = torch.arange(cout, device=DEV).view(1, -1).expand(n * h * w_, -1)
        for what, src in (("row%256", (rows % 256).float()), ("cout", cols.float())):
            r = src.reshape(n, h, w_, cout).to(BF)
            y = ops.conv2d_nhwc(x, ops.pack_conv_weight(w), b, r, ks=ks, out_f32=True).float().reshape(-1, cout)
            bad = (y != r.float().reshape(-1, cout)).nonzero()
            print(mode, what, "bad", len(bad))
            for k in range(0, min(len(bad), 200), 25):
                i, c = bad[k].tolist()
                print(f"   at row {i} (row%256 {i%256}) cout {c}: got {y[i,c].item()} expected {r.float().reshape(-1,cout)[i,c].item()}")
    else:
        r = torch.zeros(n, h, w_, cout, device=DEV).to(BF)
        y = ops.conv2d_nhwc(x, ops.pack_conv_weight(w), b, r, ks=ks, out_f32=True).float().reshape(-1, cout)
        y0 = ops.conv2d_nhwc(x, ops.pack_conv_weight(w), b, None, ks=ks, out_f32=True).float().reshape(-1, cout)
        bad = (y != y0).nonzero()
        print(mode, "bad", len(bad))
        for k in range(0, min(len(bad), 200), 25):
            i, c = bad[k].tolist()
            m = (y0 == y[i, c]).nonzero()[:3].tolist()
            print(f"   at row {i} cout {c}: got {y[i,c].item():.5f} expected {y0[i,c].item():.5f}; the value got sits at {m}")
