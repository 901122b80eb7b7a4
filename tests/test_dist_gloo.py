"""CPU, world_size 2 over gloo: the bucketed flat-gradient all-reduce averages gradients exactly like DDP would."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from dmvae_amd import dist
    from dmvae_amd.optim import FlatParams
    dist.init_distributed_mode(backend="gloo")
    assert dist.initialized() and dist.get_world_size() == world and dist.get_rank() == rank
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(16, 64), torch.nn.SiLU(), torch.nn.Linear(64, 64), torch.nn.SiLU(), torch.nn.Linear(64, 4))
    params = list(reversed(list(net.parameters())))          # backward-completion order
    fp = FlatParams(params, with_ema=False)
    sync = dist.FlatGradSync(params, fp.grad, fp.offsets, bucket_bytes=8 << 10)   # several buckets
    assert len(sync.buckets) >= 2
    results = []
    for it in range(2):
        x = torch.randn(8, 16, generator=torch.Generator().manual_seed(100 * it + rank))
        net(x).pow(2).mean().backward()
        sync.wait()
        results.append(fp.grad.clone())
        fp.zero_grad()
    # reference: average of both ranks' local gradients
    refs = []
    for it in range(2):
        acc = torch.zeros_like(fp.grad)
        for r in range(world):
            for p in net.parameters():
                p.grad = None
            x = torch.randn(8, 16, generator=torch.Generator().manual_seed(100 * it + r))
            gs = torch.autograd.grad(net(x).pow(2).mean(), params)
            for g, off in zip(gs, fp.offsets):
                acc[off:off + g.numel()] += g.reshape(-1) / world
        refs.append(acc)
    ok = all(torch.allclose(a, b, rtol=1e-5, atol=1e-7) for a, b in zip(results, refs))
    # bucket bookkeeping: contiguous, covering the whole buffer in order; every bucket's collective was started from a gradient hook (during backward), none in wait()
    b = sync.buckets
    ok = ok and b[0][0] == 0 and all(b[i][1] <= b[i + 1][0] for i in range(len(b) - 1)) and b[-1][1] == fp.offsets[-1] + params[-1].numel()
    ok = ok and sum(x[2] for x in b) == len(params) and sync.last_hook_launches == len(b)
    stats = sync.comm_stats()
    ok = ok and stats["buckets"] == len(b) and stats["launched_in_backward"] == len(b) and stats["bytes"] == b[-1][1] * 4
    dist.barrier()
    q.put((rank, ok))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_flat_grad_sync_two_ranks_gloo(world):
    """world 2, 4 and 8 (the node sizes the reference's launch scripts use: scripts/train_tokenizer.sh:20-38): averaging, bucket boundaries, every collective started
    from a gradient hook."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(world))
    assert got == [(r, True) for r in range(world)]


def test_single_process_is_noop():
    from dmvae_amd import dist
    assert not dist.initialized() or dist.get_world_size() >= 1
    t = torch.ones(3)
    assert dist.allreduce(t) is None or True
    dist.barrier()


def _bn_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from dmvae_amd import dist
    from dmvae_amd.models.patchgan import sync_batch_stats
    dist.init_distributed_mode(backend="gloo")
    sizes = [5, 9]                                           # ranks hold different batch sizes
    xs = [torch.randn(sizes[r], 16, 6, 7, generator=torch.Generator().manual_seed(7 + r)) * (1 + r) + r for r in range(world)]
    x = xs[rank]
    mean, var = x.mean(dim=(0, 2, 3)), x.var(dim=(0, 2, 3), unbiased=False)
    g_mean, g_var, n = sync_batch_stats(mean, var, x.numel() // 16)
    allx = torch.cat(xs, dim=0)
    ok = (n == allx.numel() // 16 and torch.allclose(g_mean, allx.mean(dim=(0, 2, 3)), rtol=1e-5, atol=1e-6)
          and torch.allclose(g_var, allx.var(dim=(0, 2, 3), unbiased=False), rtol=1e-4, atol=1e-6))
    dist.barrier()
    q.put((rank, ok))
    torch.distributed.destroy_process_group()


def test_sync_batchnorm_statistics_two_ranks_gloo():
    """The cross-rank combination used by the discriminator's SyncBatchNorm (models/patchgan.py:113-115 in the reference) equals the
    statistics of the concatenated batch, with unequal per-rank batch sizes."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bn_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def _bcast_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from dmvae_amd import dist
    from dmvae_amd.optim import FlatParams
    # built BEFORE init_distributed_mode(): must refuse instead of silently training unsynchronised
    torch.manual_seed(1000 + rank)                       # every rank builds DIFFERENT weights
    net = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.BatchNorm1d(32), torch.nn.Linear(32, 4))
    frozen = torch.nn.Linear(5, 5).requires_grad_(False)
    net[1].running_mean.add_(rank + 1.0)
    refused = False
    try:
        fp0 = FlatParams(list(net.parameters()), with_ema=True)
        dist.FlatGradSync(fp0.params, fp0.grad, fp0.offsets)
    except RuntimeError as e:
        refused = "init_distributed_mode" in str(e)
    dist.init_distributed_mode(backend="gloo")
    fp = FlatParams(list(net.parameters()), with_ema=True)
    extra_state = torch.full((7,), float(rank + 3))
    n = dist.broadcast_module_state(net, frozen, extra=[fp.flat, fp.ema, extra_state])
    fp.after_external_update()
    # after the broadcast every rank holds rank 0's values: gather and compare bit for bit
    mine = torch.cat([fp.flat, fp.ema, extra_state, net[1].running_mean, net[1].running_var, frozen.weight.reshape(-1), frozen.bias,
                      torch.cat([p.detach().reshape(-1) for p in net.parameters()])])
    both = [torch.empty_like(mine) for _ in range(world)]
    torch.distributed.all_gather(both, mine)
    same = all(torch.equal(both[0], b) for b in both)
    torch.manual_seed(1000)                              # rank 0's construction
    ref = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.BatchNorm1d(32), torch.nn.Linear(32, 4))
    is_rank0 = all(torch.equal(a.detach(), b.detach()) for a, b in zip(net.parameters(), ref.parameters())) and \
        torch.equal(net[1].running_mean, torch.ones(32)) and torch.equal(extra_state, torch.full((7,), 3.0))
    # parameters are still views of the flat buffer (the broadcast wrote through them, it did not re-home them)
    views = all(p.data_ptr() == fp.flat.data_ptr() + 4 * off for p, off in zip(fp.params, fp.offsets))
    dist.barrier()
    q.put((rank, refused and same and is_rank0 and views and n >= 1))
    torch.distributed.destroy_process_group()


def test_initial_state_broadcast_and_refusal_two_ranks_gloo():
    """DDP's constructor-time broadcast (reference train_tokenizer.py:302,319): ranks that built their modules from different seeds hold rank 0's
    parameters, buffers, EMA and extra optimiser state afterwards; FlatGradSync built under WORLD_SIZE > 1 before init_distributed_mode() raises."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bcast_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_bench_refuses_to_measure_fewer_gpus_than_asked():
    """`python bench.py --gpus 2` started plainly must launch its own ranks or fail loudly -- never print an n_gpus: 1 line (round-1 verdict).  This
    container has no GPU, so it has to refuse; on a 1-GPU box the same holds (tests/test_gpu_dist.py)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("this node really has two GPUs")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "--gpus 2" in r.stderr and "GPU(s)" in r.stderr
    assert '"n_gpus"' not in r.stdout
    # a launcher whose WORLD_SIZE disagrees with --gpus is refused too
    env2 = dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"], env=env2,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout) and '"n_gpus"' not in r.stdout


def _rng_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from dmvae_amd import dist, train
    dist.init_distributed_mode(backend="gloo")
    torch.manual_seed(1234 + 10000 * rank)                    # the reference's per-rank seeding (train_dmd.py:140-146)
    torch.rand(3)
    gathered = train._rng_state(all_ranks=True)               # collective: every rank's generators, keyed by rank
    single = train._rng_state()                               # this rank's only (what a master-only checkpoint holds)
    want = torch.rand(5)                                      # what the uninterrupted run draws next on THIS rank
    # resume from the gathered entry: every rank gets its own stream back
    torch.manual_seed(999)
    ok_g = train._set_rng_state(gathered) and torch.equal(torch.rand(5), want)
    # resume from rank 0's single entry: rank 0 is restored, rank 1 must NOT be given rank 0's generators
    obj = [single]
    torch.distributed.broadcast_object_list(obj, src=0)
    torch.manual_seed(4321 + 10000 * rank)
    before = torch.get_rng_state().clone()
    installed = train._set_rng_state(obj[0])
    ok_s = (installed and torch.equal(torch.rand(5), want)) if rank == 0 else (not installed and torch.equal(torch.get_rng_state(), before))
    # a checkpoint gathered at another world size restores nobody
    other = dict(gathered, world=world + 1)
    ok_w = not train._set_rng_state(other)
    dist.barrier()
    q.put((rank, bool(ok_g and ok_s and ok_w), want.tolist()))
    torch.distributed.destroy_process_group()


def test_checkpoint_rng_is_per_rank_two_ranks_gloo():
    """a checkpoint's `rng` entry must give each rank ITS generators back (or leave it alone), never rank 0's to everybody."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rng_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res
    assert res[0][2] != res[1][2], "the two ranks drew the same numbers"


def test_checkpoint_rng_single_process_round_trip():
    from dmvae_amd import train
    torch.manual_seed(7)
    st = train._rng_state()
    want = torch.rand(4)
    torch.manual_seed(8)
    assert train._set_rng_state(st) and torch.equal(torch.rand(4), want)
    legacy = {"cpu": st["cpu"]}                               # round 3's unlabelled entry: a single-process checkpoint
    torch.manual_seed(9)
    assert train._set_rng_state(legacy) and torch.equal(torch.rand(4), want)


def _eval_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from dmvae_amd import dist, evaluate as E
    from oracle.capture_golden_eval import StandInVAE, batches
    dist.init_distributed_mode(backend="gloo")
    vae = StandInVAE(5)
    data = batches(6, [2, 3, 2, 1])                           # the evaluation set: four batches, eight images
    mine = data[rank::world]                                  # each rank evaluates its shard (DistributedSampler-style)
    r = E.evaluate(vae, mine, num_samples=8)
    dist.barrier()
    q.put((rank, r["PSNR"], r["latent_mean"], r["latent_scale"], r["batches"]))
    torch.distributed.destroy_process_group()


def test_evaluate_all_reduces_like_the_reference_two_ranks_gloo():
    """train_tokenizer.py:357: psnr, latent_mean, latent_scale and the batch count are summed over the ranks before the divisions -- two ranks over two
    shards must report what one process reports over the whole set."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from dmvae_amd import evaluate as E
    from oracle.capture_golden_eval import StandInVAE, batches
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_eval_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    whole = E.evaluate(StandInVAE(5), batches(6, [2, 3, 2, 1]), num_samples=8)
    for _, ps, lm, ls, nb in res:
        assert nb == 4
        assert abs(ps - whole["PSNR"]) < 1e-9 * abs(whole["PSNR"]) + 1e-12 and abs(lm - whole["latent_mean"]) < 1e-12 and abs(ls - whole["latent_scale"]) < 1e-9 * whole["latent_scale"]
