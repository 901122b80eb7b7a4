import sys, os, time, torch
root = os.environ.get('GRAFT_REPO_ROOT', '/root/repo'); sys.path.insert(0, root)
from dmvae_amd.train import build_tokenizer_trainer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
tr = build_tokenizer_trainer()
print("trainable params", tr.fp.numel, "mem GB", torch.cuda.memory_allocated()/2**30)
g = torch.Generator(device='cuda').manual_seed(42)
images = torch.rand(B,3,256,256, device='cuda', generator=g)*2-1
for i in range(3):
    loss = tr.step(images); print(i, tr.read_log())
torch.cuda.synchronize(); t0=time.time()
N=5
for i in range(N): tr.step(images)
torch.cuda.synchronize(); dt=(time.time()-t0)/N
print(f"B={B}: {dt*1e3:.1f} ms/step  {B/dt:.1f} img/s  peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GB")
print(tr.read_log())
