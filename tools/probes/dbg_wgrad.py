import sys; sys.path.insert(0,".")
import torch, torch.nn.functional as F
from dmvae_amd import ops
for (n,h,w_,cin,cout,ks) in [(1,128,128,128,128,3),(2,64,64,256,256,3),(1,96,96,256,128,3)]:
    g=torch.Generator().manual_seed(7+cin)
    a=torch.randn(n,h,w_,cin,generator=g).cuda().bfloat16(); dy=torch.randn(n,h,w_,cout,generator=g).cuda().bfloat16()
    dw,db=ops.conv2d_nhwc_wgrad(dy,a,ks)
    ref=dy.double().sum(dim=(0,1,2))
    err=(db.double()-ref).abs()
    xr=a.double().permute(0,3,1,2); wr=torch.zeros(cout,cin,ks,ks,dtype=torch.double,device="cuda",requires_grad=True)
    F.conv2d(xr,wr,None,padding=ks//2).backward(dy.double().permute(0,3,1,2))
    print((n,h,w_,cin,cout), "db err", (err.max()/ref.abs().max()).item(), "bad", (err>1e-3*ref.abs().max()).nonzero().flatten().tolist()[:20], "dw err", ((dw.double()-wr.grad).abs().max()/wr.grad.abs().max()).item())
    print("   ratio", (db.double()/ref)[:12].tolist())
