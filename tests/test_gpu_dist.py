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


def _trainer(bucket_bytes, seed=5):
    from dmvae_amd.models.vae import VAE
    from dmvae_amd.train import TokenizerTrainer
    torch.manual_seed(seed)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        vae = VAE(z_channels=32, model_size="base", encoder_kwargs=dict(embed_dim=256, depth=1, num_heads=4)).cuda()
    return TokenizerTrainer(vae, None, warmup_steps=2, bucket_bytes=bucket_bytes)


def _worker(rank, world, port, q):
    try:
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        import torch.distributed as tdist
        from dmvae_amd import dist
        dist.init_distributed_mode(backend="gloo")
        assert dist.initialized() and dist.get_world_size() == world and torch.cuda.current_device() == 0
        # each rank builds its model from a DIFFERENT seed: the constructor-time broadcast (DDP's, train_tokenizer.py:302) must leave rank 0's
        # weights, EMA and frozen-encoder parameters everywhere
        tr = _trainer(bucket_bytes=32 << 20, seed=5 + 17 * rank)               # 200 MB of gradients: 7 buckets
        assert tr.sync.enabled and len(tr.sync.buckets) >= 4
        start = torch.cat([tr.fp.flat, tr.fp.ema, torch.cat([p.detach().reshape(-1) for p in tr.vae.encoder.parameters()])])
        starts = [torch.empty_like(start) for _ in range(world)]
        tdist.all_gather(starts, start)
        assert torch.equal(starts[0], starts[1]), "initial broadcast did not equalise the ranks"
        del start, starts
        local = _trainer(bucket_bytes=32 << 20, seed=5)            # rank 0 builds the same model again, the others receive it; gradient sync switched off below: this rank's own gradient
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
        # checkpoint() the way the reference writes one -- from the master ALONE (`if dist.is_master()`, train_tokenizer.py:439): the default must not be a
        # collective (it was, and this call would hang rank 0 in an all_gather_object), and the next collective must still line up
        ck = [tr.checkpoint() if rank == 0 else None]
        tdist.broadcast_object_list(ck, src=0)
        assert "ranks" not in ck[0]["rng"] and ck[0]["rng"]["rank"] == 0
        tr.step(x)
        tr.load(ck[0])                                      # every rank resumes from the master's file: weights / EMA / optimiser back to the step-4 state
        flats = [torch.empty_like(tr.fp.flat) for _ in range(world)]
        tdist.all_gather(flats, tr.fp.flat)
        same = same and torch.equal(flats[0], flats[1]) and tr.global_step == 4
        gathered = tr.checkpoint(all_ranks_rng=True)        # the opt-in collective form: every rank calls it
        same = same and sorted(gathered["rng"]["ranks"]) == [0, 1]
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


def test_bench_gpus_2_refused_or_run_on_this_box():
    """`python bench.py --gpus 2` started plainly: on a 1-GPU box it must exit non-zero with a clear message (never an `n_gpus: 1` line); on a
    node with >= 2 devices it launches two RCCL ranks itself and rank 0 prints `n_gpus: 2`."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900)
    if torch.cuda.device_count() >= 2:
        assert r.returncode == 0, r.stderr[-2000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        out = json.loads(line)
        assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 64 and "world=2" in r.stderr
    else:
        assert r.returncode != 0
        assert "--gpus 2" in r.stderr and "only 1 GPU" in r.stderr
        assert '"n_gpus"' not in r.stdout


def _rccl_worker(rank, world, port, q):
    try:
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                          HSA_ENABLE_IPC_MODE_LEGACY="0")
        import torch.distributed as tdist
        from dmvae_amd import dist
        dist.init_distributed_mode()                       # backend "nccl" = RCCL, one device per rank
        assert tdist.get_backend() == "nccl" and torch.cuda.current_device() == rank
        tr = _trainer(bucket_bytes=32 << 20, seed=5 + 17 * rank)
        local = _trainer(bucket_bytes=32 << 20, seed=5)
        local.sync.remove()
        local.sync.enabled = False
        assert torch.equal(tr.fp.flat, local.fp.flat)
        x = torch.rand(2, 3, 256, 256, device="cuda", generator=torch.Generator(device="cuda").manual_seed(100 + rank)) * 2 - 1
        tr.step(x)
        local.step(x)
        mine = local.fp.grad.clone()
        both = [torch.empty_like(mine) for _ in range(world)]
        tdist.all_gather(both, mine)
        ref = both[0] / world + both[1] / world
        err = ((tr.fp.grad - ref).abs().max() / ref.abs().max()).item()
        hooks_ok = tr.sync.last_hook_launches == len(tr.sync.buckets)
        for _ in range(3):
            tr.step(x)
        flats = [torch.empty_like(tr.fp.flat) for _ in range(world)]
        tdist.all_gather(flats, tr.fp.flat)
        same = torch.equal(flats[0], flats[1]) and hooks_ok
        dist.barrier()
        q.put((rank, err, same, ""))
        tdist.destroy_process_group()
    except Exception as e:
        import traceback
        q.put((rank, float("inf"), False, traceback.format_exc()[-1500:]))
        raise e


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two MI355X on one node (RCCL over xGMI)")
def test_tokenizer_trainer_two_gpus_rccl():
    """The same step on TWO devices over RCCL: bucketed asynchronous all-reduce from the gradient hooks, ranks bit-identical afterwards."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rccl_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=900) for _ in procs)
    for p in procs:
        p.join(timeout=120)
    for rank, err, same, tb in res:
        assert tb == "", tb
        assert err < 1e-6, (rank, err)
        assert same, rank


def test_bench_json_is_the_last_stdout_line_under_rccl():
    """The driver reads ONE JSON line from `bench.py`.  Under RCCL the library prints a version banner through C stdio, which (stdout being a pipe) used to come out
    at process exit -- AFTER the JSON line.  One rank with a real RCCL process group (DMVAE_FORCE_DIST=1: the collective path on a single GPU): the last line of
    stdout must parse, and carry the `comm` dictionary of an N > 1 line (bucket count, bytes, collectives launched during backward, exposed wait)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", DMVAE_FORCE_DIST="1")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, "\n".join(l for l in r.stderr.splitlines() if not l.startswith("frame #"))[:3000]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    d = json.loads(lines[-1])                                     # the LAST line, whatever native libraries printed before it
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["metric"].startswith("images/sec")
    c = d["comm"]
    assert c["backend"] == "nccl" and c["buckets"] >= 3 and c["bytes"] > 200e6 and c["launched_in_backward"] == c["buckets"] and c["wait_ms"] is not None


def test_bench_line_labels_the_dynamic_instantiation_when_more_than_one_rank_would_run():
    """What the first real multi-GPU run will print, exercised on one GPU: the collective path (DMVAE_FORCE_DIST=1: a real RCCL group of one rank) WITH dynamic tile
    claiming on (what dist.init_distributed_mode switches on for world > 1).  `roofline.kernel` must name the DYN instantiation of the dominant conv kernel and
    carry launches (a label that matches no launched kernel would report zeros), the `comm` dictionary must show every bucket but at most the last launched from a
    gradient hook during backward, an exposed wait, and `dynamic_tile_claiming`; `env` and the three windows are in the line."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", DMVAE_FORCE_DIST="1", DMVAE_PP_DYNAMIC="1")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--time-every", "1"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, "\n".join(l for l in r.stderr.splitlines() if not l.startswith("frame #"))[:3000]
    d = json.loads([l for l in r.stdout.splitlines() if l.strip()][-1])
    rf = d["roofline"]
    # template arguments <TM, TP, WM, WP, NBUF, UPS, F32, KO, GEN, SUB, DYN, STATS, HALO>: DYN is the third from the end
    args = rf["kernel"].split("<")[1].split(">")[0].replace(" ", "").split(",")
    assert args[-3] == "true", rf["kernel"]
    assert rf["launches"] > 0 and rf["achieved"] > 0 and 0 < rf["frac"] < 1, rf
    assert d["roofline_wgrad"]["launches"] > 0 and d["linear_gemm"]["launches"] > 0
    c = d["comm"]
    assert c["backend"] == "nccl" and c["dynamic_tile_claiming"] is True
    assert c["buckets"] >= 3 and c["launched_in_backward"] >= c["buckets"] - 1 and c["wait_ms"] is not None and c["wait_ms"] >= 0
    assert len(d["ms_per_step_windows"]) == 3 and d["ms_per_step_min"] <= d["ms_per_step_median"] <= d["ms_per_step_max"]
    assert set(d["env"]) >= {"sclk_mhz_avg", "power_w_avg", "temp_c", "gpu_uuid_hash"}


def _dmd_trainer(seed):
    from dmvae_amd.models.lightningdit import LightningDiT
    from dmvae_amd.models.vae import VAE
    from dmvae_amd.train import DMDTrainer
    torch.manual_seed(seed)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        vae = VAE(z_channels=32, model_size="base", encoder_kwargs=dict(embed_dim=256, depth=1, num_heads=4)).cuda()

    def dit(s):
        torch.manual_seed(s)
        m = LightningDiT(input_size=16, patch_size=1, in_channels=32, hidden_size=192, depth=4, num_heads=3, num_classes=10)
        with torch.no_grad():
            for p in m.parameters():
                if p.abs().max() == 0:
                    p.normal_(0, 0.02)
        return m.cuda()
    teacher, student = dit(seed + 1).eval().requires_grad_(False), dit(seed + 2)
    return DMDTrainer(vae, None, teacher, student, dmd_weight=5.0, dmd_cfg_scale=2.0, num_classes=10, vae_train_every=2, warmup_steps=1, bucket_bytes=1 << 20)


def _dmd_worker(rank, world, port, q):
    try:
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        import torch.distributed as tdist
        from dmvae_amd import dist
        from dmvae_amd.models import lightningdit_fast as lf
        dist.init_distributed_mode(backend="gloo")
        tr = _dmd_trainer(seed=11 + 5 * rank)                  # different seeds per rank: the constructor-time broadcast equalises VAE, student, teacher
        assert tr.sync.enabled and tr.ssync.enabled and len(tr.ssync.buckets) >= 3 and lf.STACK_SEGMENTS_DP > 1
        cat = lambda: torch.cat([tr.fp.flat, tr.sfp.flat, torch.cat([p.detach().reshape(-1) for p in tr.teacher.parameters()])])
        both = [torch.empty_like(cat()) for _ in range(world)]
        tdist.all_gather(both, cat())
        assert torch.equal(both[0], both[1]), "initial broadcast did not equalise the ranks"
        x = torch.rand(2, 3, 256, 256, device="cuda", generator=torch.Generator(device="cuda").manual_seed(100 + rank)) * 2 - 1
        y = torch.tensor([3, 7], device="cuda") + rank
        torch.manual_seed(1000 + rank)                          # per-rank noise / timestep draws, like the reference's seed + 10000 rank
        hooks = []
        for step in range(4):                                  # two cycles of (VAE turn, student-only step)
            tr.step(x, y)
            hooks.append((tr.ssync.last_hook_launches, len(tr.ssync.buckets), tr.sync.last_hook_launches if step % 2 == 0 else None, len(tr.sync.buckets)))
        tr.wait_optimizers()
        flats = [torch.empty_like(cat()) for _ in range(world)]
        tdist.all_gather(flats, cat())
        same = torch.equal(flats[0], flats[1])
        moved = not torch.equal(flats[0], both[0])
        # the student's gradient buckets were started from hooks while its backward pass was still running: all but (at most) the segment that finishes last
        early = all(h[0] >= h[1] - 1 for h in hooks)
        log = tr.read_log()
        finite = all(v == v and abs(v) < 1e6 for v in log.values())
        dist.barrier()
        q.put((rank, same and moved and early and finite, f"same {same} moved {moved} hooks {hooks} log {log}", ""))
        tdist.destroy_process_group()
    except Exception as e:
        import traceback
        q.put((rank, False, "", traceback.format_exc()[-2000:]))
        raise e


def test_dmd_trainer_two_ranks_share_one_gpu():
    """train_dmd.py's stage under data parallelism (train_dmd.py:348,355: vae_ddp and sit_ddp): BOTH FlatGradSync instances (the VAE's, the student's), the
    turn pattern, the student's block stack cut into segments so that its buckets' all-reduces start during its backward pass -- two gloo ranks on cuda:0 with
    different images / labels / noise end every cycle with bit-identical VAE, student and teacher weights."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dmd_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=900) for _ in procs)
    for p in procs:
        p.join(timeout=120)
    for rank, ok, info, tb in res:
        assert tb == "", tb
        assert ok, (rank, info)
