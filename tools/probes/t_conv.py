import sys, time, torch, ctypes
import os; sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from dmvae_amd import _lib
L = _lib.lib()
dev = 'cuda'
torch.manual_seed(0)
def run(N,H,W,Cin,Cout,ks,ups=0,bias=True,res=True,act=0,f32=False):
    x = torch.randn(N,H,W,Cin, device=dev).bfloat16()
    w = (torch.randn(Cout,Cin,ks,ks, device=dev)*0.05).bfloat16()
    b = torch.randn(Cout, device=dev) if bias else None
    Ho,Wo = (2*H,2*W) if ups else (H,W)
    r = torch.randn(N,Ho,Wo,Cout, device=dev).bfloat16() if res else None
    wp = w.permute(0,2,3,1).contiguous()  # [Cout, kh, kw, Cin]
    y = torch.empty(N,Ho,Wo,Cout, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
    d = _lib.ConvDesc(N,H,W,Cin,Cout,ks,ups,act,1 if f32 else 0)
    st = torch.cuda.current_stream().cuda_stream
    rc = L.dmvae_conv2d_nhwc_fwd(x.data_ptr(), wp.data_ptr(), b.data_ptr() if bias else None, r.data_ptr() if res else None, y.data_ptr(), ctypes.byref(d), st)
    _lib.check(rc, 'conv')
    torch.cuda.synchronize()
    # reference on CPU in fp64 from the same bf16 inputs
    xr = x.float().cpu().double().permute(0,3,1,2)
    if ups: xr = torch.nn.functional.interpolate(xr, scale_factor=2.0, mode='nearest')
    yr = torch.nn.functional.conv2d(xr, w.float().cpu().double(), b.cpu().double() if bias else None, padding=ks//2)
    if res: yr = yr + r.float().cpu().double().permute(0,3,1,2)
    if act==1: yr = yr*torch.sigmoid(yr)
    if act==2: yr = yr.relu()
    yr = yr.permute(0,2,3,1)
    err = (y.float().cpu().double()-yr).abs().max().item(); sc = yr.abs().max().item()
    print(f"N{N} H{H} W{W} {Cin}->{Cout} ks{ks} ups{ups} bias{bias} res{res} act{act} f32{f32}: maxerr {err:.3e} scale {sc:.3e} rel {err/sc:.2e}")
    return err/sc
bad=0
for cfg in [ (2,8,8,64,64,3), (2,8,8,64,128,3), (1,16,16,128,64,3), (3,5,7,32,32,3), (2,8,8,64,64,1), (1,32,32,512,512,3), (2,4,4,32,96,3,1), (1,8,8,64,256,1)]:
    for f32 in (True, False):
        e = run(*cfg, f32=f32)
        if e > (2e-5 if f32 else 6e-3): bad+=1
e = run(2,8,8,64,64,3,0,False,False,1,True); bad += e>2e-5
e = run(2,8,8,64,64,3,0,True,False,2,True); bad += e>2e-5
print("BAD", bad)
# perf
def bench(N,H,W,Cin,Cout,ks=3,ups=0,iters=20):
    x = torch.randn(N,H,W,Cin, device=dev).bfloat16()
    Ho,Wo = (2*H,2*W) if ups else (H,W)
    wp = (torch.randn(Cout,ks*ks,Cin, device=dev)*0.05).bfloat16()
    y = torch.empty(N,Ho,Wo,Cout, device=dev, dtype=torch.bfloat16)
    d = _lib.ConvDesc(N,H,W,Cin,Cout,ks,ups,0,0)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3): L.dmvae_conv2d_nhwc_fwd(x.data_ptr(), wp.data_ptr(), None, None, y.data_ptr(), ctypes.byref(d), st)
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): L.dmvae_conv2d_nhwc_fwd(x.data_ptr(), wp.data_ptr(), None, None, y.data_ptr(), ctypes.byref(d), st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/iters
    fl = 2.0*N*Ho*Wo*Cout*Cin*ks*ks
    print(f"perf N{N} {H}x{W} {Cin}->{Cout} ks{ks} ups{ups}: {ms:.3f} ms  {fl/ms/1e9:.1f} TFLOP/s")
for cfg in [(32,32,32,512,512),(32,64,64,512,512),(32,128,128,256,256),(32,256,256,128,128),(32,128,128,512,256),(32,256,256,256,128),(32,64,64,512,512,3,1),(32,128,128,256,256,3,1),(32,128,128,512,256,1)]:
    bench(*cfg)
