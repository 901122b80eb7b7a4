"""wgrad_pp HALO instantiation (128 x 384 tile, three taps of one kernel row from one 34-pixel halo tile) against an fp64 reference, and its time against the
shifted-copies form (DMVAE_WGRAD_PP_HALO=0 in a second process)."""
import sys, os, torch, ctypes
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from dmvae_amd import _lib
L = _lib.lib()
dev = 'cuda'
torch.manual_seed(0)
def run(N, H, W, Cin, Cout, acc=0):
    a = torch.randn(N, H, W, Cin, device=dev).bfloat16()
    dy = torch.randn(N, H, W, Cout, device=dev).bfloat16()
    d = _lib.ConvDesc(N, H, W, Cin, Cout, 3, 0, 0, 0)
    wsb = L.dmvae_conv2d_nhwc_wgrad_workspace(ctypes.byref(d))
    ws = torch.empty(wsb, device=dev, dtype=torch.uint8)
    dw = torch.full((Cout, Cin, 3, 3), 0.5, device=dev); db = torch.full((Cout,), 0.25, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(L.dmvae_conv2d_nhwc_wgrad(dy.data_ptr(), a.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), wsb, ctypes.byref(d), acc, st), 'wgrad')
    torch.cuda.synchronize()
    xr = a.double().permute(0, 3, 1, 2).contiguous()
    wr = torch.zeros(Cout, Cin, 3, 3, dtype=torch.double, device=dev, requires_grad=True); br = torch.zeros(Cout, dtype=torch.double, device=dev, requires_grad=True)
    yr = torch.nn.functional.conv2d(xr, wr, br, padding=1)
    yr.backward(dy.double().permute(0, 3, 1, 2))
    gw, gb = wr.grad + (0.5 if acc else 0), br.grad + (0.25 if acc else 0)
    e1 = (dw.double() - gw).abs().max().item() / gw.abs().max().item()
    e2 = (db.double() - gb).abs().max().item() / gb.abs().max().item()
    print(f"wgrad N{N} {H}x{W} {Cin}->{Cout} acc{acc}: rel dw {e1:.2e} db {e2:.2e}", flush=True)
    return max(e1, e2)
def bench(N, H, W, Cin, Cout, iters=10):
    a = torch.randn(N, H, W, Cin, device=dev).bfloat16(); dy = torch.randn(N, H, W, Cout, device=dev).bfloat16()
    d = _lib.ConvDesc(N, H, W, Cin, Cout, 3, 0, 0, 0)
    wsb = L.dmvae_conv2d_nhwc_wgrad_workspace(ctypes.byref(d)); ws = torch.empty(wsb, device=dev, dtype=torch.uint8)
    dw = torch.zeros(Cout, Cin, 3, 3, device=dev); db = torch.zeros(Cout, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    f = lambda: L.dmvae_conv2d_nhwc_wgrad(dy.data_ptr(), a.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), wsb, ctypes.byref(d), 0, st)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"bench N{N} {H}x{W} {Cin}->{Cout}: {ms * 1e3:8.1f} us {2.0 * N * H * W * Cin * Cout * 9 / ms / 1e9:7.1f} TF/s", flush=True)
if len(sys.argv) > 1 and sys.argv[1] == "bench":
    for cfg in [(32, 256, 256, 128, 128), (32, 256, 256, 256, 128), (32, 128, 128, 128, 128), (64, 128, 128, 128, 128)]:
        bench(*cfg)
else:
    bad = 0
    for cfg in [(4, 32, 32, 128, 128), (2, 64, 64, 256, 128), (1, 8, 64, 128, 128), (3, 5, 128, 128, 128, 1), (1, 32, 192, 128, 256), (2, 16, 256, 256, 128), (1, 32, 64, 128, 128, 1), (2, 64, 32, 384, 128), (1, 64, 64, 128, 384), (1, 128, 128, 128, 128), (5, 32, 32, 256, 128)]:
        bad += run(*cfg) > 2e-5
    print("BAD", bad)
