"""Does running the GroupNorm backward (reduce pass + apply pass) over batch SLICES, so that the apply pass re-reads da / x from the 256-MiB Infinity Cache
instead of HBM, pay?  Times the whole-batch call against per-slice calls (slices of s samples)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dmvae_amd import ops
def timed(fn, reps=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (n, h, w, c) in [(32, 256, 256, 128), (32, 256, 256, 256), (32, 128, 128, 256), (32, 128, 128, 512), (32, 64, 64, 512)]:
    # rotate over several buffer sets so that no call finds its operands cached from the previous repetition
    nset = max(2, int(700e6 // (n * h * w * c * 2 * 3)) + 1)
    sets = []
    for _ in range(nset):
        sets.append((torch.randn(n, h, w, c, device="cuda").to(torch.bfloat16), torch.randn(n, h, w, c, device="cuda").to(torch.bfloat16), torch.randn(n, h, w, c, device="cuda").to(torch.bfloat16)))
    g, b = torch.randn(c, device="cuda"), torch.randn(c, device="cuda")
    st = ops.groupnorm_stats(sets[0][0])
    ctr = [0]
    def whole():
        ctr[0] = (ctr[0] + 1) % nset
        x, da, dres = sets[ctr[0]]
        ops.groupnorm_bwd(da, x, st, g, b, True, dres=dres)
    def sliced(s):
        def f():
            ctr[0] = (ctr[0] + 1) % nset
            x, da, dres = sets[ctr[0]]
            for i in range(0, n, s):
                ops.groupnorm_bwd(da[i:i + s], x[i:i + s], st[i:i + s], g, b, True, dres=dres[i:i + s])
        return f
    S = n * h * w * c * 2
    line = f"[{n},{h},{w},{c}] S={S/1e6:.0f} MB  whole {timed(whole):7.1f} us"
    for s in (16, 8, 4, 2):
        line += f" | s={s} {timed(sliced(s)):7.1f}"
    print(line, flush=True)
