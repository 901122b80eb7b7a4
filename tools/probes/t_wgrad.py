import sys, os, time, torch, ctypes
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from dmvae_amd import _lib
L = _lib.lib()
dev = 'cuda'
torch.manual_seed(0)
def run(N,H,W,Cin,Cout,ks,ups=0,acc=0):
    Ho,Wo = (2*H,2*W) if ups else (H,W)
    a = torch.randn(N,H,W,Cin, device=dev).bfloat16()
    dy = torch.randn(N,Ho,Wo,Cout, device=dev).bfloat16()
    d = _lib.ConvDesc(N,H,W,Cin,Cout,ks,ups,0,0)
    wsb = L.dmvae_conv2d_nhwc_wgrad_workspace(ctypes.byref(d))
    ws = torch.empty(wsb, device=dev, dtype=torch.uint8)
    dw = torch.full((Cout,Cin,ks,ks), 0.5, device=dev); db = torch.full((Cout,), 0.25, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(L.dmvae_conv2d_nhwc_wgrad(dy.data_ptr(), a.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), wsb, ctypes.byref(d), acc, st), 'wgrad')
    torch.cuda.synchronize()
    xr = a.float().cpu().double().permute(0,3,1,2).contiguous()
    if ups: xr = torch.nn.functional.interpolate(xr, scale_factor=2.0, mode='nearest')
    wr = torch.zeros(Cout,Cin,ks,ks, dtype=torch.double, requires_grad=True); br = torch.zeros(Cout, dtype=torch.double, requires_grad=True)
    yr = torch.nn.functional.conv2d(xr, wr, br, padding=ks//2)
    yr.backward(dy.float().cpu().double().permute(0,3,1,2))
    gw, gb = wr.grad + (0.5 if acc else 0), br.grad + (0.25 if acc else 0)
    e1 = (dw.cpu().double()-gw).abs().max().item()/gw.abs().max().item()
    e2 = (db.cpu().double()-gb).abs().max().item()/gb.abs().max().item()
    print(f"wgrad N{N} {H}x{W} {Cin}->{Cout} ks{ks} ups{ups} acc{acc}: rel dw {e1:.2e} db {e2:.2e}")
    return max(e1,e2)
bad=0
for cfg in [(2,8,8,64,64,3),(2,8,8,64,128,3,0,1),(1,16,16,128,64,3),(3,5,7,32,32,3),(2,8,8,64,64,1),(1,32,32,512,512,3),(2,4,4,32,96,3,1),(1,8,8,64,256,1),(2,16,16,32,512,3),(2,3,3,40,24,3)]:
    bad += run(*cfg) > 2e-5
print("BAD", bad)
def bench(N,H,W,Cin,Cout,ks=3,ups=0,iters=10):
    Ho,Wo = (2*H,2*W) if ups else (H,W)
    a = torch.randn(N,H,W,Cin, device=dev).bfloat16(); dy = torch.randn(N,Ho,Wo,Cout, device=dev).bfloat16()
    d = _lib.ConvDesc(N,H,W,Cin,Cout,ks,ups,0,0)
    wsb = L.dmvae_conv2d_nhwc_wgrad_workspace(ctypes.byref(d)); ws = torch.empty(wsb, device=dev, dtype=torch.uint8)
    dw = torch.zeros(Cout,Cin,ks,ks, device=dev); db = torch.zeros(Cout, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for withb in (0,1):
        f = lambda: L.dmvae_conv2d_nhwc_wgrad(dy.data_ptr(), a.data_ptr(), dw.data_ptr(), db.data_ptr() if withb else None, ws.data_ptr(), wsb, ctypes.byref(d), 0, st)
        for _ in range(2): f()
        torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)/iters
        fl = 2.0*N*Ho*Wo*Cout*Cin*ks*ks
        print(f"perf wgrad N{N} {H}x{W} {Cin}->{Cout} ks{ks} ups{ups} bias{withb}: {ms:.3f} ms  {fl/ms/1e9:.1f} TFLOP/s  ws {wsb/1e6:.0f} MB")
for cfg in [(32,32,32,512,512),(32,64,64,512,512),(32,128,128,256,256),(32,256,256,128,128),(32,128,128,512,256),(32,256,256,256,128),(32,64,64,512,512,3,1),(32,128,128,512,256,1)]:
    bench(*cfg)
