import sys, os, torch, ctypes
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from dmvae_amd import _lib
L = _lib.lib(); dev='cuda'; torch.manual_seed(0)
st = lambda: torch.cuda.current_stream().cuda_stream
def run(N,H,W,C,swish,G=32,res=True):
    HW=H*W
    x = (torch.randn(N,HW,C, device=dev)*1.5+0.3).bfloat16()
    gamma = torch.randn(C, device=dev)*0.5+1; beta = torch.randn(C, device=dev)*0.2
    da = torch.randn(N,HW,C, device=dev).bfloat16(); dres = torch.randn(N,HW,C, device=dev).bfloat16() if res else None
    wsb = L.dmvae_groupnorm_workspace(N,HW,C,G); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    stats = torch.empty(N,G,2, device=dev); y = torch.empty_like(x); dx = torch.empty_like(x)
    dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
    _lib.check(L.dmvae_groupnorm_stats(x.data_ptr(), stats.data_ptr(), ws.data_ptr(), wsb, N,HW,C,G,1e-6, st()),'stats')
    _lib.check(L.dmvae_groupnorm_apply(x.data_ptr(), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), N,HW,C,G,swish, st()),'apply')
    _lib.check(L.dmvae_groupnorm_bwd(da.data_ptr(), x.data_ptr(), dres.data_ptr() if res else None, stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), dx.data_ptr(), dg.data_ptr(), db.data_ptr(), ws.data_ptr(), wsb, N,HW,C,G,swish,0, st()),'bwd')
    torch.cuda.synchronize()
    xr = x.float().cpu().double().permute(0,2,1).reshape(N,C,H,W).requires_grad_(True)
    gr = gamma.cpu().double().requires_grad_(True); br = beta.cpu().double().requires_grad_(True)
    yr = torch.nn.functional.group_norm(xr, G, gr, br, 1e-6)
    if swish: yr = yr*torch.sigmoid(yr)
    yr.backward(da.float().cpu().double().permute(0,2,1).reshape(N,C,H,W))
    mean = xr.detach().reshape(N,G,-1).mean(-1); var = xr.detach().reshape(N,G,-1).var(-1, unbiased=False)
    e_m = (stats[...,0].cpu().double()-mean).abs().max().item(); e_r = ((stats[...,1].cpu().double()-1/torch.sqrt(var+1e-6)).abs()/ (1/torch.sqrt(var+1e-6))).max().item()
    yq = yr.detach().reshape(N,C,HW).permute(0,2,1)
    e_y = (y.float().cpu().double()-yq).abs().max().item()/yq.abs().max().item()
    dxr = xr.grad.reshape(N,C,HW).permute(0,2,1) + (dres.float().cpu().double() if res else 0)
    e_dx = (dx.float().cpu().double()-dxr).abs().max().item()/dxr.abs().max().item()
    e_dg = (dg.cpu().double()-gr.grad).abs().max().item()/gr.grad.abs().max().item(); e_db = (db.cpu().double()-br.grad).abs().max().item()/br.grad.abs().max().item()
    print(f"gn N{N} {H}x{W} C{C} swish{swish}: mean {e_m:.1e} rstd {e_r:.1e} y {e_y:.1e} dx {e_dx:.1e} dgamma {e_dg:.1e} dbeta {e_db:.1e}")
    return max(e_m, e_r, e_dg, e_db) > 2e-5 or max(e_y, e_dx) > 5e-3
bad=0
for cfg in [(2,8,8,64,1),(2,8,8,32,1),(1,16,16,128,0),(3,5,7,96,1),(2,32,32,512,1),(2,64,64,256,1),(1,64,64,128,1,32,False)]:
    bad += run(*cfg)
print("BAD", bad)
def bench(N,H,W,C):
    HW=H*W; G=32
    x = torch.randn(N,HW,C, device=dev).bfloat16(); gamma=torch.ones(C,device=dev); beta=torch.zeros(C,device=dev)
    da = torch.randn(N,HW,C, device=dev).bfloat16()
    wsb = L.dmvae_groupnorm_workspace(N,HW,C,G); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    stats = torch.empty(N,G,2, device=dev); y = torch.empty_like(x); dx=torch.empty_like(x); dg=torch.zeros(C,device=dev); db=torch.zeros(C,device=dev)
    fs = {'stats': lambda: L.dmvae_groupnorm_stats(x.data_ptr(), stats.data_ptr(), ws.data_ptr(), wsb, N,HW,C,G,1e-6, st()),
          'apply': lambda: L.dmvae_groupnorm_apply(x.data_ptr(), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), N,HW,C,G,1, st()),
          'bwd': lambda: L.dmvae_groupnorm_bwd(da.data_ptr(), x.data_ptr(), None, stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), dx.data_ptr(), dg.data_ptr(), db.data_ptr(), ws.data_ptr(), wsb, N,HW,C,G,1,0, st())}
    byt = {'stats':2,'apply':4,'bwd':10}
    for k,f in fs.items():
        for _ in range(2): f()
        torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize(); ms=e0.elapsed_time(e1)/10
        print(f"perf gn {k} N{N} {H}x{W} C{C}: {ms:.3f} ms  {x.numel()*byt[k]/ms/1e6:.0f} GB/s")
bench(32,256,256,128); bench(32,128,128,256); bench(32,32,32,512)
