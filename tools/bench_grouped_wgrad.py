"""The grouped Linear weight-gradient launch at LightningDiT-XL/1's backward shapes (4 x 28 problems; B = 16 / 64 -> M = 4096 / 16384 token rows) and at a ViT-L block's
(4 problems, M = 4112): ms per launch and TFLOP/s.  DMVAE_WGRAD_GROUPED_XCD=0 runs the first placement (every problem spread over all XCDs) for an A/B."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dmvae_amd import ops
BF = torch.bfloat16
print("DMVAE_WGRAD_GROUPED_XCD =", os.environ.get("DMVAE_WGRAD_GROUPED_XCD", "1 (default)"))
for name, m, layers, shapes in (("DiT-XL/1 B=16", 4096, 28, [(3456, 1152), (1152, 1152), (6144, 1152), (1152, 3072)]), ("DiT-XL/1 B=64", 16384, 28, [(3456, 1152), (1152, 1152), (6144, 1152), (1152, 3072)]),
                                ("ViT-L block B=16", 4112, 1, [(3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096)])):
    xs = {c: torch.randn(m, c, device="cuda").to(BF) for c in {s[0] for s in shapes} | {s[1] for s in shapes}}
    probs = []
    for _ in range(layers):
        for cout, cin in shapes:
            probs.append((xs[cout], xs[cin], torch.empty(cout, cin, device="cuda"), None))
    # one activation buffer per width is shared by the layers (memory): a cache-friendlier case than the step's distinct tensors at B = 16, the same at B = 64 (operands >> L2)
    for _ in range(3):
        ops.linear_wgrad_grouped(probs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n):
        ops.linear_wgrad_grouped(probs)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    fl = layers * sum(2 * m * a * b for a, b in shapes)
    print(f"{name:18s} {len(probs):4d} problems  {ms:7.3f} ms  {fl / ms * 1e-9:6.0f} TFLOP/s")
    del probs, xs
