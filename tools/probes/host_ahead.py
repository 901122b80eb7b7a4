"""How far the host runs ahead of the device in the DMD stage (C3) and the diffusion step (C4): wall time of ENQUEUEING a step (no synchronisation) against the
device time of the same steps.  If enqueueing takes nearly as long as the device needs, launch gaps are not hidden and a captured graph would pay."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dmvae_amd.train import build_dmd_trainer, build_diffusion_trainer
dev = torch.device("cuda:0")
for name, build, B, warm in (("dmd", build_dmd_trainer, 16, 5), ("diffusion", build_diffusion_trainer, 64, 3)):
    tr = build(device=dev)
    images = torch.rand(B, 3, 256, 256, device=dev, generator=torch.Generator(device=dev).manual_seed(1)) * 2 - 1
    labels = torch.randint(0, 1000, (B,), device=dev, generator=torch.Generator(device=dev).manual_seed(2))
    for _ in range(warm):
        tr.step(images, labels)
    torch.cuda.synchronize()
    host = []
    t0 = time.perf_counter()
    for _ in range(10):
        a = time.perf_counter()
        kind = "turn" if name == "dmd" and tr.global_step % tr.vae_train_every == 0 else "step"
        tr.step(images, labels)
        host.append((kind, (time.perf_counter() - a) * 1e3))
    t_enq = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) * 1e3
    for k in sorted({k for k, _ in host}):
        v = [x for kk, x in host if kk == k]
        print(f"{name:10s} {k}: host enqueue {sum(v) / len(v):7.2f} ms (min {min(v):.2f}, max {max(v):.2f}) over {len(v)} steps")
    print(f"{name:10s} 10 steps: enqueued after {t_enq:.1f} ms, device done after {t_all:.1f} ms")
    del tr, images, labels
    import gc; gc.collect(); torch.cuda.empty_cache()
