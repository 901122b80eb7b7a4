"""The data-parallel step on real kernels with TWO processes: both ranks share cuda:0 (the GPU box has one device; RCCL refuses two ranks on one
device, so the collective runs over gloo on device tensors -- the transport is not what is under test).  What is: `dist.FlatGradSync` driven by the
post-accumulate hooks of parameters whose gradients the HIP backward Functions write DIRECTLY into the flat buffer, the asynchronous bucket launches
interleaved with the remaining backward kernels, `wait()` before the fused optimiser step, and that ranks starting from the same weights with different
images stay bit-identical."""
import os
import socket
import warnings

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _trainer(bucket_bytes):
    from dmvae_amd.models.vae import VAE
    from dmvae_amd.train import TokenizerTrainer
    torch.manual_seed(5)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        vae = VAE(z_channels=32, model_size="base", encoder_kwargs=dict(embed_dim=64, depth=1, num_heads=2)).cuda()
    return TokenizerTrainer(vae, None, warmup_steps=2, bucket_bytes=bucket_bytes)


def _worker(rank, world, port, q):
    try:
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        import torch.distributed as tdist
        from dmvae_amd import dist
        dist.init_distributed_mode(backend="gloo")
        assert dist.initialized() and dist.get_world_size() == world and torch.cuda.current_device() == 0
        tr = _trainer(bucket_bytes=32 << 20)               # 200 MB of gradients: 7 buckets
        assert tr.sync.enabled and len(tr.sync.buckets) >= 4
        local = _trainer(bucket_bytes=32 << 20)            # same weights, gradient sync switched off: this rank's own gradient
        local.sync.remove()
        local.sync.enabled = False
        assert torch.equal(tr.fp.flat, local.fp.flat)
        x = torch.rand(2, 3, 256, 256, device="cuda", generator=torch.Generator(device="cuda").manual_seed(100 + rank)) * 2 - 1
        tr.step(x)
        local.step(x)
        mine = local.fp.grad.clone()
        both = [torch.empty_like(mine) for _ in range(world)]
        tdist.all_gather(both, mine)
        assert not torch.equal(both[0], both[1])           # the ranks really saw different images
        ref = both[0] / world + both[1] / world
        err = ((tr.fp.grad - ref).abs().max() / ref.abs().max()).item()
        # every bucket was started from a gradient hook, i.e. while backward was still running (the direct flat-buffer gradient writes still go through
        # AccumulateGrad, so the post-accumulate hooks fire); nothing was left for wait() to launch
        assert tr.sync.last_hook_launches == len(tr.sync.buckets), (tr.sync.last_hook_launches, len(tr.sync.buckets))
        for _ in range(3):                                  # lr warm-up: the weights move from the second step on
            tr.step(x)
        flats = [torch.empty_like(tr.fp.flat) for _ in range(world)]
        tdist.all_gather(flats, tr.fp.flat)
        same = torch.equal(flats[0], flats[1]) and not torch.equal(tr.fp.flat, local.fp.flat)
        dist.barrier()
        q.put((rank, err, same, ""))
        tdist.destroy_process_group()
    except Exception as e:          # surface the failure instead of a bare exit code
        import traceback
        q.put((rank, float("inf"), False, traceback.format_exc()[-1500:]))
        raise e


def test_tokenizer_trainer_two_ranks_share_one_gpu():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
    for rank, err, same, tb in res:
        assert tb == "", tb
        assert err < 1e-6, (rank, err)      # (g0 / 2 + g1 / 2) in f32 on both sides
        assert same, rank
