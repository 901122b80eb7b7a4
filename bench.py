#!/usr/bin/env python
"""Benchmark: images/sec of one DMVAE tokenizer train step @256x256 (BASELINE.json metric) on N MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1 started plainly (no RANK in the env): this process checks that N devices exist and re-executes itself as N ranks, one per GPU, through
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` (the reference's launcher: scripts/train_tokenizer.sh:20-38);
started BY torch.distributed.run it is one of those ranks.  Either way WORLD_SIZE must equal --gpus and every rank must own a distinct device, or the
run exits non-zero -- it never falls back to measuring one GPU.

Step = VAE forward (frozen ViT-L/16 encoder, bottleneck MLP, conv decoder) + L1 + LPIPS + backward + bucketed RCCL
gradient all-reduce + clip + AdamW + EMA, bf16 compute, local batch 32 (train_tokenizer.py, config C2 of SURVEY.md 8),
synthetic images, random-init weights of the reference architecture (no network for data / checkpoints).
Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (implicit-GEMM conv forward/dgrad, csrc/conv_pp.hip, MFMA-bound),
timed with HIP events on its launch stream inside the timed region; `cpu_baseline` is the CPU oracle
(oracle/ref_cpu.py, "port") running the same whole step in fp32 at batch 2 on the host cores -- all usable physical cores and 8 threads (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # this pool's host driver only supports dmabuf IPC; RCCL needs it for N > 1 (already exported on the GPU boxes)

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0      # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
LOCAL_BATCH = 32


def host_cores():
    """(os.cpu_count(), usable logical CPUs, physical cores among them).  Usable = scheduler affinity capped by the cgroup CPU quota:
    a container that sees 256 CPUs but may run on 16 thrashes when handed 256 threads (round 1 measured 357 s for a step that takes 20 s)."""
    logical = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = logical
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                txt = f.read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    usable = min(usable, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f2:
                        usable = min(usable, max(1, q // int(f2.read())))
            break
        except (OSError, ValueError, IndexError):
            continue
    phys = set()
    try:
        pid = cid = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    pid = line.split(":")[1].strip()
                elif line.startswith("core id"):
                    cid = line.split(":")[1].strip()
                elif not line.strip():
                    if pid is not None and cid is not None:
                        phys.add((pid, cid))
                    pid = cid = None
    except OSError:
        pass
    physical = len(phys) if phys else max(1, logical // 2)
    return logical, usable, max(1, min(physical, usable))


class GpuEnvSampler:
    """Clock / power / temperature of the device during the timed region, sampled from sysfs by a side thread (no subprocess, nothing on the stream): the MFMA
    kernels run at whatever clock the package's power and thermal state allows, so two boxes -- or one box at two times -- differ by 2-4 % on the same build;
    with these in the line a 2 % difference between two bench lines can be attributed.  Every field is null where the node does not expose it."""

    def __init__(self, device_index: int = 0, period_s: float = 0.05):
        import glob
        import threading
        self.period, self.samples, self._stop, self._thread = period_s, {"sclk_mhz": [], "power_w": [], "temp_c": []}, threading.Event(), None
        self.card, self.hwmon, self.uid = None, None, None
        # the device's sysfs node by its PCI address (a node exposes one drm card per GPU / partition of the whole host: the visible device is not card0)
        try:
            pr = torch.cuda.get_device_properties(device_index)
            bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
            node = os.path.join("/sys/bus/pci/devices", bdf)
            if os.path.isdir(node):
                self.card = node
        except (AttributeError, RuntimeError, AssertionError):
            pass
        if self.card is None:       # no PCI address from the runtime: the only AMD card, if there is exactly one
            cards = []
            for c in sorted(glob.glob("/sys/class/drm/card[0-9]*")):
                try:
                    with open(os.path.join(c, "device", "vendor")) as f:
                        if f.read().strip() == "0x1002":
                            cards.append(os.path.join(c, "device"))
                except OSError:
                    continue
            if len(cards) == 1:
                self.card = cards[0]
        if self.card:
            hw = sorted(glob.glob(os.path.join(self.card, "hwmon", "hwmon*")))
            self.hwmon = hw[0] if hw else None
            try:
                import hashlib
                with open(os.path.join(self.card, "unique_id")) as f:
                    self.uid = hashlib.sha1(f.read().strip().encode()).hexdigest()[:10]
            except OSError:
                pass

    @staticmethod
    def _num(path):
        try:
            with open(path) as f:
                return float(f.read().split()[0])
        except (OSError, ValueError, IndexError):
            return None

    def _sclk(self):
        v = self._num(os.path.join(self.hwmon, "freq1_input")) if self.hwmon else None
        if v is not None:
            return v / 1e6
        try:                                    # pp_dpm_sclk: "N: 2100Mhz *" marks the current level
            with open(os.path.join(self.card, "pp_dpm_sclk")) as f:
                for ln in f:
                    if "*" in ln:
                        return float(ln.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
        except (OSError, ValueError, IndexError, TypeError):
            pass
        return None

    def _run(self):
        while not self._stop.is_set():
            if self.card:
                s = self._sclk()
                p = None
                if self.hwmon:
                    p = self._num(os.path.join(self.hwmon, "power1_average")) or self._num(os.path.join(self.hwmon, "power1_input"))
                t = None
                if self.hwmon:
                    for name in ("temp2_input", "temp1_input"):        # junction where there is one, else edge
                        t = self._num(os.path.join(self.hwmon, name))
                        if t is not None:
                            break
                for k, v in (("sclk_mhz", s), ("power_w", None if p is None else p / 1e6), ("temp_c", None if t is None else t / 1e3)):
                    if v is not None:
                        self.samples[k].append(v)
            self._stop.wait(self.period)

    def start(self):
        import threading
        self._thread = threading.Thread(target=self._run, daemon=True)
        self._thread.start()

    def stop(self) -> dict:
        self._stop.set()
        if self._thread is not None:
            self._thread.join(timeout=2.0)
        avg = lambda v: round(sum(v) / len(v), 1) if v else None
        s = self.samples
        return {"sclk_mhz_avg": avg(s["sclk_mhz"]), "sclk_mhz_min": round(min(s["sclk_mhz"]), 1) if s["sclk_mhz"] else None, "power_w_avg": avg(s["power_w"]),
                "power_w_max": round(max(s["power_w"]), 1) if s["power_w"] else None, "temp_c": avg(s["temp_c"]), "gpu_uuid_hash": self.uid,
                "samples": len(s["sclk_mhz"]) or len(s["power_w"]) or len(s["temp_c"]), "sysfs_node": self.card, "source": "sysfs hwmon of the device, sampled every %d ms by a side thread during the timed region" % int(self.period * 1e3)}


def cpu_baseline_worker(threads: int, batch: int) -> None:
    """Child process of `cpu_baseline`: ONE whole fp32 tokenizer train step of the CPU oracle (oracle/ref_cpu.py::tokenizer_train_steps -- VAE forward with
    the frozen ViT-L encoder, L1 + MSE + LPIPS, backward, clip_grad_norm_, AdamW, EMA; train_tokenizer.py:403-437) at batch `batch` on `threads` threads.
    SURVEY.md 8d / BASELINE.md 3: images = rand(B,3,256,256, seed 42) * 2 - 1, weights from torch.manual_seed(42) + the reference's constructor order."""
    import warnings
    torch.set_num_threads(threads)
    from oracle import ref_cpu as R
    from dmvae_amd.models.vae import VAE
    from dmvae_amd.utils.lpips import LPIPS
    torch.manual_seed(42)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        vae = VAE(z_channels=32, model_size="large")
        lp = LPIPS()
    p = {k: v.detach() for k, v in vae.state_dict().items()}
    trainable = [k for k in p if k.startswith(("decoder.", "bottle_neck."))]
    lp_p = {k: v.detach() for k, v in lp.state_dict().items()}
    x = torch.rand(batch, 3, 256, 256, generator=torch.Generator().manual_seed(42)) * 2 - 1
    t0 = time.time()
    logs, _, _ = R.tokenizer_train_steps(x, p, lp_p, trainable, steps=1, num_heads=16)
    dt = time.time() - t0
    print("CPU_BASELINE " + json.dumps({"threads": threads, "batch": batch, "seconds": round(dt, 2), "images_per_sec": round(batch / dt, 4),
                                        "rec_loss": round(logs[0]["rec_loss"], 5)}), flush=True)


def cpu_baseline(batch=2, timeout_s=150.0):
    """the CPU oracle ("port") running the SAME step (fp32, no autocast, batch 2, full step including clip + AdamW + EMA) on this
    box's host cores, once on all usable physical cores and once on 8 threads; each in its own process (fresh thread pools) under a timeout so that
    the default bench run stays within minutes.  `value` is the faster of the runs that finished."""
    import subprocess
    logical, usable, physical = host_cores()
    runs = []
    for threads in sorted({physical, min(8, usable)}, reverse=True):
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", str(threads), "--batch", str(batch)]
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
        t0 = time.time()
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout_s)
            line = [l for l in r.stdout.splitlines() if l.startswith("CPU_BASELINE ")]
            if r.returncode == 0 and line:
                runs.append(dict(json.loads(line[-1][len("CPU_BASELINE "):]), finished=True))
            else:
                runs.append({"threads": threads, "batch": batch, "finished": False, "error": (r.stderr or r.stdout)[-300:]})
        except subprocess.TimeoutExpired:
            runs.append({"threads": threads, "batch": batch, "finished": False, "timeout_s": timeout_s,
                         "images_per_sec_upper_bound": round(batch / (time.time() - t0), 4)})
    done = [r for r in runs if r.get("finished")]
    best = max(done, key=lambda r: r["images_per_sec"]) if done else None
    return {"value": best["images_per_sec"] if best else None, "unit": "images/sec", "cores": best["threads"] if best else None, "kind": "port",
            "os_cpu_count": logical, "usable_cpus": usable, "physical_cores_used_for_all_cores_run": physical, "runs": runs,
            "sample": f"1 whole tokenizer train step at batch {batch} (fp32, no autocast; oracle/ref_cpu.py::tokenizer_train_steps: ViT-L fwd + MLP + decoder "
                      f"fwd/bwd + L1 + MSE + LPIPS + clip_grad_norm + AdamW + EMA), timed once per thread count in a fresh process "
                      f"(model construction excluded); value = the faster finished run"}


def kl_mmd_roofline(dev):
    """The build-defined KL + MMD op, timed with HIP events on its launch stream:
    (1) the fused call at the step's real shape (B=32 images x 256 latent tokens x 32 channels vs 256 prior samples);
    (2) the HBM-bound part alone -- the KL moment pass + its gradient -- on a tensor larger than the 256 MB Infinity Cache;
    (3) the pairwise (VALU/exp-bound) kernel at a large batch.  Algorithmic bytes: z and y read once, dz written once (fused call);
    z read by the moment pass, z read + dz written by the gradient pass (KL-only)."""
    from dmvae_amd import ops
    HBM_PEAK = 8000.0

    def timed(fn, reps):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3     # us

    def timed_graph(fn, reps):
        """The same `reps` calls recorded once into a HIP graph and replayed: what the op costs on its stream when the host is not the limit (issuing the
        two launches from Python takes longer than they run; inside the training step the host runs ahead of the stream anyway)."""
        # Not next to a live process group: stream capture in its default (global) mode makes EVERY thread's event queries illegal while it lasts, and RCCL's
        # watchdog thread polls its work events all the time -- "HIP error: operation not permitted when stream is capturing" in the watchdog, process abort
        # (seen once per few dozen runs of the one-rank RCCL test; with N ranks each of them rolls that die).  There the eager figure is reported.
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            return None
        try:
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):      # other threads (the env sampler, a profiler) keep their API calls
                for _ in range(reps):
                    fn()
            g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps * 1e3
        except Exception:      # capture unsupported: report the eager figure only
            return None

    out = {}
    z = torch.randn(32, 256, 32, device=dev) * 0.7 + 0.2
    y = torch.randn(32, 256, 32, device=dev)
    us_eager = timed(lambda: ops.kl_mmd(z, y, need_grad=True), 50)
    us_graph = timed_graph(lambda: ops.kl_mmd(z, y, need_grad=True), 20)
    us = us_graph if us_graph is not None else us_eager
    byt = 3 * z.numel() * 4
    pairs = 32 * 3 * 256 * 256
    # scalar pair kernel (large batches): per kernel evaluation with gradient 32 FMA (a.b) + 32 FMA (sum w b) + ~14 (norm combine, five bandwidths by repeated
    # squaring, weights) + one v_exp_f32 (quarter rate: 4 issue slots) = ~82 f32 lane-operations; peak = 157.3 TFLOP/s / 2 = 78.6 T lane-FMA/s (MI355X_MICROARCH.md)
    LANE_OPS_PER_PAIR, VALU_PEAK_TOPS = 82.0, 78.65
    out["fused_B32"] = {"shape": "G=32 n=m=256 d=32, value+grad", "us_per_call": round(us, 1), "timing": "20 calls replayed from one HIP graph" if us_graph is not None else "eager",
                        "us_per_call_issued_from_python": round(us_eager, 1), "launches": 2, "algorithmic_MB": round(byt / 1e6, 2),
                        "achieved_GBps": round(byt / us / 1e3, 1), "hbm_frac": round(byt / us / 1e3 / HBM_PEAK, 4),
                        "Gpair_per_s": round(pairs / us / 1e3, 1),
                        # this shape runs the matrix-core pair kernel: 64 FLOP (a.b) per pair + 64 FLOP (sum w b) for the two thirds of the pairs that carry a gradient
                        "mfma_f32_TFLOPs": round(pairs * (64.0 + 64.0 * 2 / 3) / us / 1e6, 2), "mfma_f32_frac": round(pairs * (64.0 + 64.0 * 2 / 3) / us / 1e6 / 157.3, 4),
                        "bound": "latency (two launches of ~5 us floor each) + f32 matrix cores / exp, not HBM (3 MB of traffic against 6.3 M kernel evaluations): see DESIGN_HISTORY.md 3.4"}
    zl = torch.randn(8192, 256, 32, device=dev)       # 268 MB
    us = timed(lambda: ops.kl_mmd(zl, None, need_grad=True), 10)
    byt = 3 * zl.numel() * 4
    out["kl_pass_268MB"] = {"shape": "G=8192 n=256 d=32, KL moments + gradient only", "us_per_call": round(us, 1),
                            "algorithmic_MB": round(byt / 1e6, 1), "achieved_GBps": round(byt / us / 1e3, 1),
                            "hbm_frac": round(byt / us / 1e3 / HBM_PEAK, 4), "bound": "hbm"}
    del zl
    zb = torch.randn(1024, 256, 32, device=dev)
    yb = torch.randn(1024, 256, 32, device=dev)
    us = timed(lambda: ops.kl_mmd(zb, yb, need_grad=True), 5)
    byt = 3 * zb.numel() * 4
    pairs = 1024 * 3 * 256 * 256
    out["fused_B1024"] = {"shape": "G=1024 n=m=256 d=32, value+grad", "us_per_call": round(us, 1), "algorithmic_MB": round(byt / 1e6, 1),
                          "achieved_GBps": round(byt / us / 1e3, 1), "hbm_frac": round(byt / us / 1e3 / HBM_PEAK, 4),
                          "Gpair_per_s": round(pairs / us / 1e3, 1), "valu_Tlaneops_per_s": round(pairs * LANE_OPS_PER_PAIR / us / 1e6, 2),
                          "valu_frac": round(pairs * LANE_OPS_PER_PAIR / us / 1e6 / VALU_PEAK_TOPS, 4), "bound": "valu/exp"}
    return out


# Work per image of the secondary configurations (GFLOP, forward; BASELINE.md section 2 -- measured with hooks on the reference's modules): backward = 2 x forward
_GF = {"vit": 155.6 + 6.5, "mlp": 1.1, "dec": 620.3 + 2.1, "vgg": 40.1, "dit": 228.8 + 8.5}


def secondary_stages(dev, which=("c3", "c4", "gan")) -> dict:
    """The other stages of the reference's recipe at their script configurations, timed on this device AFTER the C2 headline (same process, rank 0, N = 1; the
    headline `value` stays C2): `c3_dmd_cycle` = train_dmd.py's step (config C3: local batch 16, ViT-L/16 trainable, LightningDiT-XL/1 teacher + student, CFG 5,
    the VAE on every 5th step: train.build_dmd_trainer = what tests/test_gpu_fullsize.py::test_dmd_stage_full_size_cycle_c3 builds) over two 5-step cycles;
    `c4_diffusion_step` = train_diffusion.py's step (config C4: local batch 64, frozen VAE, LightningDiT-XL/1, AdamW + EMA); `gan_step` = the tokenizer step from
    `disc_start_step` on (generator's adaptive-weight GAN term + the PatchGAN discriminator's hinge / BCR step).  Wall clock between device synchronisations;
    `frac_of_peak` = algorithmic TFLOP per step (BASELINE.md section 2 x batch; backward = 2 x forward) / time / 2.5 PFLOP/s."""
    import gc
    from dmvae_amd.train import build_diffusion_trainer, build_dmd_trainer, build_tokenizer_trainer
    out = {}

    def timed(fn, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    def free():
        gc.collect()
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()

    if "c3" in which:
        free()
        B = 16
        tr = build_dmd_trainer(device=dev)
        images = torch.rand(B, 3, 256, 256, device=dev, generator=torch.Generator(device=dev).manual_seed(1)) * 2 - 1
        labels = torch.randint(0, 1000, (B,), device=dev, generator=torch.Generator(device=dev).manual_seed(2))
        for _ in range(5):
            tr.step(images, labels)          # one whole cycle: warm
        ms = timed(lambda: tr.step(images, labels), 10)      # two whole cycles between two synchronisations: the figure a training run sees
        turn, stud = [], []
        for _ in range(10):                  # two more cycles, every step between its own synchronisations: the split by step kind
            (turn if tr.global_step % tr.vae_train_every == 0 else stud).append(timed(lambda: tr.step(images, labels), 1))
        tf_student = B * (_GF["vit"] + _GF["mlp"] + 3 * _GF["dit"]) / 1e3
        tf_turn = B * (3 * _GF["vit"] + 3 * _GF["mlp"] + 3 * _GF["dec"] + 3 * _GF["vgg"] + 4 * _GF["dit"] + 3 * _GF["dit"]) / 1e3
        tf_cycle = (tf_turn + 4 * tf_student) / 5
        log = tr.read_log()
        out["c3_dmd_cycle"] = {
            "workload": "train_dmd.py step (C3): VAE(large, z 32) with the ViT-L/16 encoder trainable + LPIPS + DMD loss (LightningDiT-XL/1 teacher + student, CFG 5, cond + uncond as one 2B call) "
                        "every 5th step; the student's flow-matching step (forward + backward + clip + AdamW on 675 M parameters) every step",
            "local_batch": B, "steps_timed": 10, "ms_per_step": round(ms, 2), "timing": "ms_per_step: ten consecutive steps (two cycles) between two device synchronisations; vae_turn_ms / student_ms: ten further steps, each between its own", "vae_turn_ms": round(sum(turn) / len(turn), 2), "student_ms": round(sum(stud) / len(stud), 2),
            "images_per_sec": round(B / ms * 1e3, 1), "tflop_per_step": round(tf_cycle, 2), "tflop_vae_turn": round(tf_turn, 2), "tflop_student_step": round(tf_student, 2),
            "frac_of_peak": round(tf_cycle / ms / MFMA_BF16_PEAK_TFLOPS * 1e3, 4), "frac_of_peak_student_step": round(tf_student / (sum(stud) / len(stud)) / MFMA_BF16_PEAK_TFLOPS * 1e3, 4),
            "frac_of_peak_vae_turn": round(tf_turn / (sum(turn) / len(turn)) / MFMA_BF16_PEAK_TFLOPS * 1e3, 4),
            "peak_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1), "dtype": "bf16", "data": "synthetic",
            "log_after_run": {k: round(v, 5) for k, v in log.items() if k in ("rec_loss", "dmd_loss", "diffusion_loss")}}
        del tr, images, labels
    if "c4" in which:
        free()
        B = 64
        tr = build_diffusion_trainer(device=dev)
        images = torch.rand(B, 3, 256, 256, device=dev, generator=torch.Generator(device=dev).manual_seed(1)) * 2 - 1
        labels = torch.randint(0, 1000, (B,), device=dev, generator=torch.Generator(device=dev).manual_seed(2))
        for _ in range(3):
            tr.step(images, labels)
        ms = timed(lambda: tr.step(images, labels), 8)
        tf = B * (_GF["vit"] + _GF["mlp"] + 3 * _GF["dit"]) / 1e3
        out["c4_diffusion_step"] = {
            "workload": "train_diffusion.py step (C4): frozen VAE.encode (ViT-L/16 + bottleneck) -> flow-matching loss on LightningDiT-XL/1 -> clip -> AdamW -> EMA",
            "B": B, "steps_timed": 8, "ms": round(ms, 2), "images_per_sec": round(B / ms * 1e3, 1), "tflop_per_step": round(tf, 2),
            "frac": round(tf / ms / MFMA_BF16_PEAK_TFLOPS * 1e3, 4), "peak_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1), "dtype": "bf16",
            "data": "synthetic", "loss_after_run": round(tr.read_log()["loss"], 5)}
        del tr, images, labels
    if "gan" in which:
        free()
        B = LOCAL_BATCH
        tr = build_tokenizer_trainer(device=dev, seed=42, with_disc=True, disc_start_step=0)
        images = torch.rand(B, 3, 256, 256, device=dev, generator=torch.Generator(device=dev).manual_seed(42)) * 2 - 1
        for _ in range(3):
            tr.step(images)
        ms = timed(lambda: tr.step(images), 10)
        lg, dl = tr.read_log(), tr.read_disc_log()
        out["gan_step"] = {"workload": "train_tokenizer.py step from disc_start_step on: C2's step + the generator's adaptive-weight adversarial term + the PatchGAN discriminator's hinge / BCR update",
                           "local_batch": B, "steps_timed": 10, "ms": round(ms, 2), "images_per_sec": round(B / ms * 1e3, 1), "d_weight": round(lg.get("d_weight", 0.0), 5),
                           "d_loss": round(dl["d_loss"], 5)}
        del tr, images
    free()
    return out


def _latest_profile(suffix: str) -> str:
    """The newest round's `profiles/rN_<suffix>` (the PMC passes are re-collected on each round's build: the figure read back belongs to the kernels that run)."""
    import re
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    best = None
    for f in os.listdir(d):
        m = re.match(r"r(\d+)_" + re.escape(suffix) + "$", f)
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), f)
    if best is None:
        raise OSError("no profiles/rN_" + suffix)
    return best[1]


def self_launch(n_gpus: int) -> int:
    """No RANK in the env and --gpus N > 1: become the launcher.  One rank per device over RCCL, rendezvous on 127.0.0.1, a free port; the
    ranks' stdout/stderr pass straight through (rank 0 prints the JSON line).  Returns the exit code of the job."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n_gpus:
        print(f"bench.py: --gpus {n_gpus} but only {have} GPU(s) are visible on this node; refusing to measure fewer GPUs than asked for",
              file=sys.stderr, flush=True)
        return 2
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n_gpus)))
    print(f"bench.py: launching {n_gpus} ranks (one per GPU, RCCL): {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=LOCAL_BATCH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the c3_dmd_cycle / c4_diffusion_step / gan_step keys (N = 1 only; ~1.5 min of model construction and steps)")
    ap.add_argument("--secondary", default="c3,c4,gan", help="which of the secondary stage keys to measure")
    ap.add_argument("--time-every", type=int, default=4, help="record the per-launch roofline events on every n-th timed step (1 = all)")
    ap.add_argument("--cpu-baseline-worker", type=int, default=0, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        cpu_baseline_worker(args.cpu_baseline_worker, args.batch)
        return
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if "RANK" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus))

    from dmvae_amd import dist, ops
    from dmvae_amd.train import build_tokenizer_trainer
    env_world = int(os.environ.get("WORLD_SIZE", "1")) if "RANK" in os.environ else 1
    if env_world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={env_world}; one rank per GPU is the contract")
    if not torch.cuda.is_available() or torch.cuda.device_count() < (args.gpus if os.environ.get("DMVAE_DIST_BACKEND") != "gloo" else 1):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but {torch.cuda.device_count() if torch.cuda.is_available() else 0} GPU(s) visible")
    dist.init_distributed_mode()
    rank, world = dist.get_rank(), dist.get_world_size()
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the process group has {world} rank(s)")
    dev = torch.device("cuda", torch.cuda.current_device())
    if world > 1:
        # every rank owns a distinct device (RCCL needs it; a shared device would also double-count the hardware)
        ids = [None] * world
        torch.distributed.all_gather_object(ids, (os.environ.get("LOCAL_RANK"), torch.cuda.current_device()))
        if os.environ.get("DMVAE_DIST_BACKEND") != "gloo" and len({d for _, d in ids}) != world:
            raise SystemExit(f"bench.py: ranks share devices: {ids}")
        if rank == 0:
            be = torch.distributed.get_backend()
            print(f"bench.py: backend {be}{' (RCCL)' if be == 'nccl' else ''} world={world}, devices {[d for _, d in ids]}", file=sys.stderr, flush=True)

    tr = build_tokenizer_trainer(device=dev, seed=42)
    gen = torch.Generator(device=dev).manual_seed(42 + rank)
    images = torch.rand(args.batch, 3, 256, 256, device=dev, generator=gen) * 2 - 1

    for _ in range(args.warmup):
        tr.step(images)
    dist.barrier()
    torch.cuda.synchronize()
    _flush_c_stdio()      # every rank: whatever the collective library printed while its communicators came up (RCCL's version banner) leaves libc's buffer NOW, long before rank 0's JSON line
    # Per-launch HIP events (roofline) bracket every conv / weight-gradient call of the steps they are on: ~290 event records per step, each a marker packet
    # the stream has to retire (measured: ~1 ms per step).  They are recorded on every `--time-every`-th step of the timed region (default 4: the first, the
    # fifth, ...), so the figure is still measured live inside the timed region while the timed region pays a quarter of that cost; 1 = every step.
    timing_all = []
    tr.sync.time_wait = True
    # The timed region is EXACTLY --steps steps between the two barrier + synchronize brackets (the contract); inside it, events on the compute stream at the
    # third-points split those same steps into three windows, so that the line also says how much the step time moved within the run (no extra steps, no sync)
    marks = sorted({0, args.steps // 3, (2 * args.steps) // 3, args.steps})
    win_ev = {}
    sampler = GpuEnvSampler(torch.cuda.current_device()) if rank == 0 else None
    if sampler is not None:
        sampler.start()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if i in marks:
            win_ev[i] = torch.cuda.Event(enable_timing=True)
            win_ev[i].record()
        ops.KERNEL_TIMING = timing_all if i % max(1, args.time_every) == 0 else None
        tr.step(images)
    win_ev[args.steps] = torch.cuda.Event(enable_timing=True)
    win_ev[args.steps].record()
    ops.KERNEL_TIMING = timing_all
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    env_stats = sampler.stop() if sampler is not None else None
    windows = [round(win_ev[a].elapsed_time(win_ev[b]) / (b - a), 3) for a, b in zip(marks, marks[1:]) if b > a]
    timing, ops.KERNEL_TIMING = timing_all, None
    tdt = torch.tensor([dt], dtype=torch.float64, device=dev)
    if dist.initialized():
        torch.distributed.all_reduce(tdt, op=torch.distributed.ReduceOp.MAX)
    dt = tdt.item()
    log = tr.read_log()
    if rank != 0:
        _emit("", rank0=False)      # leave the group together with rank 0 (it prints the line after everybody's native output is out)
        return
    frac_timed = len(range(0, args.steps, max(1, args.time_every))) / max(1, args.steps)     # share of the timed steps that carried the per-launch events
    # the plain 256x256 instantiation; its dynamic-tile-claiming twin (DYN = true) when more than one rank runs (dmvae_amd/dist.py sets DMVAE_PP_DYNAMIC)
    # (its kx-halo form, HALO = true: the 3x3 launches; the 1x1 launches of the same tile stay on the HALO = false instantiation)
    DOMINANT = "conv_pp_kernel<256, 256, 2, 4, 4, false, false, true, false, false, %s, false, %s>" % (
        "true" if os.environ.get("DMVAE_PP_DYNAMIC", "0") not in ("", "0") else "false", "true" if ops._PP_HALO else "false")
    per = {}
    for label, e0, e1, fl in timing:
        a = per.setdefault(label, [0.0, 0.0, 0])
        a[0] += e0.elapsed_time(e1); a[1] += fl; a[2] += 1
    k_ms, k_flop, k_n = per.get(DOMINANT, [0.0, 0.0, 0])
    conv = {k: v for k, v in per.items() if not k.startswith("wgrad") and k != "gemm_pp_kernel"}
    lin_ms, lin_flop, lin_n = per.get("gemm_pp_kernel", [0.0, 0.0, 0])      # the encoder's Linear layers (csrc/gemm_pp.hip): their own line below
    all_ms = sum(a[0] for a in conv.values())
    all_flop = sum(a[1] for a in conv.values())
    n_conv = sum(a[2] for a in conv.values())
    wg_ms, wg_flop, wg_n = per.get("wgrad_pp", [0.0, 0.0, 0])
    wgs_ms, wgs_flop, wgs_n = per.get("wgrad_small", [0.0, 0.0, 0])
    wg_ach = wg_flop / (wg_ms * 1e-3) / 1e12 if wg_ms > 0 else 0.0
    achieved = k_flop / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
    # HBM bytes per launch of that kernel: PMC counters cannot be read from inside this process, so the figure is the one collected over
    # this same command by tools/pmc_step_traffic.sh (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, gfx950 correction applied)
    # and committed under profiles/; null when the committed profile is for a different kernel
    traffic, traffic_src = None, None
    try:
        tname = _latest_profile("conv_pp_traffic.json")
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", tname)) as f:
            tp = json.load(f)
        static_twin = "conv_pp_kernel<256, 256, 2, 4, 4, false, false, true, false, false, false, false, %s>" % ("true" if ops._PP_HALO else "false")
        if tp.get("kernel") == static_twin and args.batch == LOCAL_BATCH:      # the DYN twin runs the same kernel body
            traffic = int(tp["hbm_MB_per_launch"] * 1e6)
            traffic_src = "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over this command in round %s, %d launches; FETCH_SIZE x2 per MI355X_MICROARCH.md)" % (
                tname, tp.get("round"), tp["launches_profiled"])
    except (OSError, ValueError, KeyError):
        pass
    wg_traffic = None
    try:
        wname = _latest_profile("wgrad_pp_traffic.json")
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", wname)) as f:
            wt = json.load(f)
        if args.batch == LOCAL_BATCH:
            # both instantiations that carry the weight-gradient time: the 256 x 256 tile (plain 3x3 / 1x1 form) and the 128 x 384 halo tile of the >= 2^19-pixel shapes
            wg_traffic = {"bytes_per_launch": int(wt["hbm_MB_per_launch"] * 1e6), "kernel": wt["kernel"], "launches_profiled": wt["launches_profiled"],
                          "per_kernel": [{"kernel": k["kernel"], "bytes_per_launch": int(k["hbm_MB_per_launch"] * 1e6), "read_MB": k["hbm_read_MB_per_launch"],
                                          "write_MB": k["hbm_write_MB_per_launch"], "launches_profiled": k["launches_profiled"]} for k in wt.get("kernels", [])],
                          "source": "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over this command, tools/pmc_traffic_r3.sh; FETCH_SIZE x2 per MI355X_MICROARCH.md)" % wname}
    except (OSError, ValueError, KeyError):
        pass
    out = {
        "metric": "images/sec DMVAE train step @256x256",
        "value": round(world * args.batch * args.steps / dt, 2),
        "unit": "images/sec",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3),
        # the same timed steps in three consecutive windows (stream events at the third-points, rank 0's device): spread within the run
        "ms_per_step_windows": windows, "ms_per_step_min": min(windows) if windows else None, "ms_per_step_max": max(windows) if windows else None,
        "ms_per_step_median": sorted(windows)[len(windows) // 2] if windows else None,
        "env": env_stats,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "train_tokenizer.py VAE pretrain step (C2): ViT-L/16 frozen encoder + MLP + flux decoder, L1 + LPIPS, "
                               "AdamW + EMA, ImageNet-256-shaped synthetic batch, random-init weights",
                   "local_batch": args.batch, "global_batch": world * args.batch, "image": "3x256x256", "z_channels": 32,
                   "parallelism": f"dp{world}", "loss_after_run": round(log["rec_loss"], 5)},
        "roofline": {"bound": "mfma", "kernel": "dmvae_conv_pp::" + DOMINANT + " (conv forward / input-gradient, Cout >= 256; decoder + LPIPS trunk)",
                     "achieved": round(achieved, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_unit": "HBM bytes per launch",
                     "traffic_source": traffic_src,
                     "launches": k_n, "launches_timed_on": "every %d-th step of the timed region" % max(1, args.time_every), "avg_launch_us": round(k_ms * 1e3 / max(1, k_n), 2),
                     "share_of_step": round(k_ms / (dt * 1e3 * frac_timed), 3),
                     "all_conv_fwd_dgrad_launches": {"launches": n_conv, "achieved_TFLOPs": round(all_flop / (all_ms * 1e-3) / 1e12, 1) if all_ms > 0 else 0.0,
                                                     "share_of_step": round(all_ms / (dt * 1e3 * frac_timed), 3)}},
        # the second MFMA-bound family: conv weight gradients (csrc/conv_wgrad_pp.hip: split-K over pixels, transpose reads, + slab reduce and
        # the bias gradient), timed per call like the forward / input-gradient launches
        "roofline_wgrad": {"bound": "mfma", "kernel": "dmvae_wgrad_pp::wgrad_pp_kernel + wgrad_reduce_kernel (conv / Linear weight + bias gradient, whole call)",
                           "achieved": round(wg_ach, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(wg_ach / MFMA_BF16_PEAK_TFLOPS, 4),
                           "traffic": wg_traffic, "launches": wg_n, "avg_launch_us": round(wg_ms * 1e3 / max(1, wg_n), 2),
                           "share_of_step": round(wg_ms / (dt * 1e3 * frac_timed), 3),
                           "small_shape_calls": {"launches": wgs_n, "share_of_step": round(wgs_ms / (dt * 1e3 * frac_timed), 3)}},
    }
    out["linear_gemm"] = {"kernel": "dmvae_gemm_pp::gemm_pp_kernel (the frozen ViT-L encoder's Linear layers: qkv / proj / fc1 + GELU / fc2, patch embedding)",
                          "launches": lin_n, "achieved_TFLOPs": round(lin_flop / (lin_ms * 1e-3) / 1e12, 1) if lin_ms > 0 else 0.0,
                          "frac": round(lin_flop / (lin_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4) if lin_ms > 0 else 0.0,
                          "avg_launch_us": round(lin_ms * 1e3 / max(1, lin_n), 2), "share_of_step": round(lin_ms / (dt * 1e3 * frac_timed), 3)}
    if world > 1 or os.environ.get("DMVAE_FORCE_DIST", "0") != "0":
        # what the gradient exchange looked like from this rank: bytes all-reduced per step, buckets, how many of their collectives were launched from
        # gradient hooks DURING backward, and the time the compute stream then still spent waiting for them (the un-overlapped remainder)
        out["comm"] = dict(tr.sync.comm_stats(), backend=str(torch.distributed.get_backend()) if dist.initialized() else None,
                           bucket_bytes=64 << 20, dynamic_tile_claiming=os.environ.get("DMVAE_PP_DYNAMIC", "0") not in ("", "0"))      # dist.init_distributed_mode sets it when it switches DYN on
    if world == 1:
        out["kl_mmd"] = kl_mmd_roofline(dev)
    del tr, images
    if world == 1 and not args.no_secondary and os.environ.get("DMVAE_FORCE_DIST", "0") == "0":
        torch.cuda.empty_cache()
        out["comm_n1"] = comm_n1_overhead(out["ms_per_step"])
    torch.cuda.empty_cache()
    if world == 1 and not args.no_secondary:
        try:
            out.update(secondary_stages(dev, tuple(x for x in args.secondary.split(",") if x)))
        except Exception as e:      # noqa: BLE001 -- the headline line must still come out; the failure is reported in it
            out["secondary_error"] = f"{type(e).__name__}: {e}"[:400]
    if world == 1 and not args.no_cpu_baseline:
        torch.cuda.empty_cache()
        out["cpu_baseline"] = cpu_baseline()
    _emit(json.dumps(out), rank0=True)


def _flush_c_stdio() -> None:
    """RCCL prints its version banner through C stdio; with stdout a pipe that text sits in libc's buffer until the process exits -- i.e. it would come out AFTER the
    JSON line, which has to be the last thing this command prints.  Flushing libc's streams puts it where it was written."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass


def comm_n1_overhead(headline_ms: float) -> dict:
    """What the data-parallel machinery costs BEFORE any byte moves between GPUs, measured on this one GPU (no multi-GPU node was available to any round; the first
    real SCALE line has this term of DESIGN.md 4's budget to be held against): the same C2 step in a fresh process with a real RCCL group of ONE rank
    (DMVAE_FORCE_DIST=1: post-accumulate hooks, four bucketed all-reduces with ncclAvg, wait() before the optimiser) and dynamic tile claiming in the persistent
    conv kernels (DMVAE_PP_DYNAMIC=1: what dist.init_distributed_mode switches on under DP, because a collective's resident kernel takes CUs).  Also records which
    RCCL the box runs and whether NCCL_ALGO / NCCL_PROTO pin the all-reduce's algorithm (unset = RCCL's tuner decides per message size)."""
    import re
    import socket
    import subprocess
    s_ = socket.socket()
    s_.bind(("127.0.0.1", 0))
    port = s_.getsockname()[1]
    s_.close()
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), DMVAE_FORCE_DIST="1", DMVAE_PP_DYNAMIC="1",
               NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT,COLL,TUNING")
    res = {"what": "C2 step with a one-rank RCCL group + dynamic tile claiming, fresh process, 10 steps after 3", "headline_ms": round(headline_ms, 3)}
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-secondary"], env=env,
                           capture_output=True, text=True, timeout=240)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not lines:
            res["error"] = (r.stderr or r.stdout)[-300:]
            return res
        sub = json.loads(lines[-1])
        res["forced_ms"] = sub["ms_per_step"]
        res["overhead_ms"] = round(sub["ms_per_step"] - headline_ms, 3)
        res["comm"] = sub.get("comm")
        info = [l for l in (r.stdout + "\n" + r.stderr).splitlines() if "NCCL INFO" in l]
        pick = [re.sub(r"^.*NCCL INFO ", "", l)[:160] for l in info if re.search(r"(?i)\b(algo|proto|ring|tree|channel|version|AllReduce)", l)]
        res["rccl_info"] = pick[:6]
    except Exception as e:      # noqa: BLE001 -- a diagnostic key must not cost the headline line
        res["error"] = f"{type(e).__name__}: {e}"[:300]
    try:
        res["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:            # noqa: BLE001
        res["rccl_version"] = None
    res["NCCL_ALGO"], res["NCCL_PROTO"] = os.environ.get("NCCL_ALGO", "unset (tuner)"), os.environ.get("NCCL_PROTO", "unset (tuner)")
    return res


def _emit(line: str, rank0: bool) -> None:
    """The ONE JSON line, as the last line on stdout.  With a process group: every rank finishes its device work, meets the others, flushes whatever native
    libraries buffered (RCCL's banner), rank 0 prints, and the process LEAVES WITHOUT TEARING THE GROUP DOWN (`os._exit`): `destroy_process_group()` / interpreter
    shutdown race ProcessGroupNCCL's watchdog thread, which then aborts the process (rc -6, seen twice in ~12 runs of this command under a real RCCL group in round 4:
    `c10d::ProcessGroupNCCL::Watchdog::run` rethrowing a HIP error from an event query) -- after the measurement, but before / instead of the line.  Nothing is lost
    by not running destructors: the line is out, the OS reclaims the rest."""
    grouped = torch.distributed.is_available() and torch.distributed.is_initialized()
    if grouped:
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        torch.distributed.barrier()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
    _flush_c_stdio()
    sys.stdout.flush()
    if rank0:
        print(line, flush=True)
    if grouped:
        sys.stderr.flush()
        # Under a profiler (rocprofv3 / roctracer / torch.profiler flush their output from exit hooks) or with DMVAE_BENCH_SOFT_EXIT=1: leave the group the
        # ordinary way -- destroy_process_group(), then interpreter shutdown -- so the trace is written; the line is already out.  Otherwise the hard exit.
        prof = any(k.startswith(("ROCPROF", "ROCP_", "ROCTRACER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
        if prof or os.environ.get("DMVAE_BENCH_SOFT_EXIT", "0") not in ("", "0"):
            try:
                torch.distributed.destroy_process_group()
            except Exception as e:      # noqa: BLE001 -- the measurement is out; report and leave
                print(f"bench.py: destroy_process_group failed after the line was printed: {e}", file=sys.stderr, flush=True)
            return
        os._exit(0)


if __name__ == "__main__":
    main()
