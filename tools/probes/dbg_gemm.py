import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dmvae_amd import _lib, ops
L = ctypes.CDLL(_lib.LIB_PATH); L.dmvae_debug_gemm_cfg.argtypes = [ctypes.c_int]
g = torch.Generator(device="cuda").manual_seed(0)
for m, n, k in [(8192, 1024, 768), (8224, 1024, 1024), (777, 520, 192), (4096, 1152, 1152)]:
    x = torch.randn(m, k, device="cuda", generator=g).to(torch.bfloat16); w = (torch.randn(n, k, device="cuda", generator=g) * k ** -0.5).to(torch.bfloat16); b = torch.randn(n, device="cuda", generator=g)
    ref = x.double() @ w.double().t()
    for cfg in range(10):
        L.dmvae_debug_gemm_cfg(cfg)
        errs = []
        for bias in (b, b.to(torch.bfloat16), None, b, None):
            y = ops.linear_bf16(x, w, bias, out_f32=True)
            r = ref + (bias.double() if bias is not None else 0)
            errs.append(((y.double() - r).abs().max() / r.abs().max()).item())
        print(m, n, k, "cfg", cfg, " ".join("%.1e" % e for e in errs), flush=True)
