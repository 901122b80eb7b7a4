import torch, time, sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch.nn.functional as F
def bench(tag):
    for (n, k) in [(3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096)]:
        x = torch.randn(32, 257, k, device='cuda').bfloat16(); w = torch.randn(n, k, device='cuda').bfloat16(); b = torch.randn(n, device='cuda').bfloat16()
        for _ in range(5): F.linear(x, w, b)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): F.linear(x, w, b)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        print(tag, n, k, f"{us:.1f} us {2*8224*n*k/us/1e6:.0f} TF/s")
bench("default")
from dmvae_amd import gemm_select
print("enable ->", gemm_select.enable(), torch.cuda.tunable.is_enabled(), torch.cuda.tunable.tuning_is_enabled())
print(torch.cuda.tunable.get_results())
bench("table")
