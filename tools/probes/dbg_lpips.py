import sys; sys.path.insert(0,"."); sys.path.insert(0,"tests")
import torch
from dmvae_amd import ops
from test_gpu_lpips import _lpips
lp=_lpips().cuda()
g=torch.Generator().manual_seed(1)
img=(torch.rand(2,3,64,64,generator=g)*2-1).cuda()
x=torch.cat([img,img],0)
h=ops.nchw_to_nhwc_bf16(((x-lp.scaling_layer.shift)/lp.scaling_layer.scale).contiguous(), c_pad=32)
print("in eq", torch.equal(h[:2],h[2:]))
from dmvae_amd import functional as Fn
import torch.nn as nn
convs=[m for sl in (lp.net.slice1,lp.net.slice2,lp.net.slice3,lp.net.slice4,lp.net.slice5) for m in sl if isinstance(m,nn.Conv2d)]
from dmvae_amd.utils.lpips import _CFG
ci=0
for v in _CFG:
    if v=="M":
        h=ops.maxpool2x2(h); print("pool eq", torch.equal(h[:2],h[2:])); continue
    c=convs[ci]; wp=Fn.packed(c.weight,False,0,32 if c.weight.shape[1]<32 else 0)
    h=ops.conv2d_nhwc(h,wp,c.bias.detach().float(),ks=3,act=2)
    d=(h[:2].float()-h[2:].float()).abs().max().item()
    print("conv",ci,tuple(h.shape),"eq",torch.equal(h[:2],h[2:]),"maxdiff",d)
    if ci in (1,3,6,9,12):
        out=torch.zeros(1,device="cuda")
        n,hh,ww,_=h.shape
        ops.lpips_diff(h[:2].contiguous(),h[2:].contiguous(),torch.ones(h.shape[-1],device="cuda"),out,1.0/(hh*ww*2),False,accumulate=False)
        print("   lpips level value", out.item())
    ci+=1
