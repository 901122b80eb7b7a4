"""One launch rewrites every packed weight operand of a flat parameter buffer after the optimiser step (dmvae_pack_weights_batched, functional.repack_all):
bit-identical to the per-weight pack launches (dmvae_pack_conv_weight_v2 / dmvae_subpixel_weight) it replaces, in place, with the cache entries left current."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _params():
    g = torch.Generator().manual_seed(5)
    shapes = [(64, 32, 3, 3), (32, 64, 1, 1), (128, 128, 3, 3), (96, 40, 3, 3), (64, 64, 3, 3), (256, 128), (3, 128, 3, 3)]
    return [torch.nn.Parameter((torch.randn(*s, generator=g) * 0.1).to(DEV)) for s in shapes]


def _requests(ps):
    """(parameter, packed() keyword arguments): forward / input-gradient operands, padded ones, Upsample's sub-pixel pair, a Linear weight with its K-tile-major copy."""
    return [(ps[0], {}), (ps[0], dict(for_dgrad=True)), (ps[1], {}), (ps[1], dict(for_dgrad=True)), (ps[2], {}), (ps[2], dict(for_dgrad=True)),
            (ps[3], dict(rows_pad=128, cols_pad=64)), (ps[3], dict(for_dgrad=True, rows_pad=64, cols_pad=128)),
            (ps[4], dict(subpixel=True)), (ps[4], dict(for_dgrad=True, subpixel=True)), (ps[5], dict(kmajor=True)), (ps[6], dict(rows_pad=32))]


def _fresh(w, for_dgrad=False, rows_pad=0, cols_pad=0, subpixel=False, kmajor=False):
    """The operand(s) by plain torch indexing -- independent of both pack kernels: out[co][t][ci] = w[co][ci][t]; for_dgrad: out[ci][T-1-t][co] = w[co][ci][t];
    sub-pixel: WD[ci][co][r][s] = sum of the taps of W[co][ci] landing on source pixel (r, s), added in (ky, kx) order; K-tile-major copy [cols/32][T][rows][32]."""
    src = w.detach().float()
    if subpixel:
        cout, cin = src.shape[:2]
        wd = torch.zeros(cin, cout, 4, 4, device=src.device)
        for r in range(4):
            for s_ in range(4):
                acc = torch.zeros(cout, cin, device=src.device)
                for ky in range(3):
                    for kx in range(3):
                        if 2 <= r + ky <= 3 and 2 <= s_ + kx <= 3:
                            acc = acc + src[:, :, ky, kx]
                wd[:, :, r, s_] = acc.t()
        src = wd
    if src.dim() == 2:
        src = src[:, :, None, None]
    co, ci, ks = src.shape[0], src.shape[1], src.shape[2]
    T = ks * ks
    m = src.reshape(co, ci, T)
    out = m.permute(1, 2, 0).flip(1) if for_dgrad else m.permute(0, 2, 1)          # [rows][T][cols]
    rows, cols = out.shape[0], out.shape[2]
    rp, cp = max(rows_pad, rows), max(cols_pad, cols)
    full = torch.zeros(rp, T, cp, device=src.device)
    full[:rows, :, :cols] = out
    full = full.to(torch.bfloat16)
    want_km = (w.dim() == 4 and ks in (3, 4)) or (kmajor and w.dim() == 2)      # functional.packed: 3x3 / 4x4 operands (the sub-pixel one is 4x4) and frozen Linear weights
    if want_km and cp % 32 == 0:
        full._dmvae_kmajor = full.view(rp, T, cp // 32, 32).permute(2, 1, 0, 3).contiguous()
    return full


def test_single_weight_pack_matches_torch_indexing():
    """dmvae_pack_conv_weight_v2 (the element-wise kernel) against plain torch indexing, every request shape of this file incl. padding; the batched test below
    holds the tiled kernel of the one-launch table to the same reference."""
    from dmvae_amd import ops
    ps = _params()
    for w, kw in _requests(ps):
        if kw.get("subpixel"):
            continue
        got = ops.pack_conv_weight(w.detach(), kw.get("for_dgrad", False), kw.get("rows_pad", 0), kw.get("cols_pad", 0),
                                   kmajor=(w.dim() == 4 and w.shape[2] == 3) or (kw.get("kmajor", False) and w.dim() == 2))
        want = _fresh(w, **kw)
        assert got.shape == want.shape and torch.equal(got, want), (tuple(w.shape), kw)
        if hasattr(want, "_dmvae_kmajor"):
            assert torch.equal(got._dmvae_kmajor, want._dmvae_kmajor), (tuple(w.shape), kw)
    w4 = torch.randn(48, 40, 4, 4, device=DEV)            # 16 taps (PatchGAN's 4x4 convs)
    for fd in (False, True):
        assert torch.equal(ops.pack_conv_weight(w4, fd), _fresh(w4, for_dgrad=fd))


def test_batched_repack_matches_per_weight_packs(monkeypatch):
    from dmvae_amd import functional as Fn
    from dmvae_amd.optim import FlatAdamWEMA, FlatParams
    monkeypatch.setattr(Fn, "_PACK_BATCHED", True)
    ps = _params()
    fp = FlatParams(ps, with_ema=False)
    opt = FlatAdamWEMA(fp, lr=1e-2, warmup_steps=0, max_norm=0.0)
    reqs = _requests(ps)
    first = [Fn.packed(w, **kw) for w, kw in reqs]
    assert len(fp.pack_reg["entries"]) == len(reqs)
    for step in range(3):
        fp.grad.copy_(torch.randn(fp.numel, device=DEV, generator=torch.Generator(device=DEV).manual_seed(step)))
        before = [p.clone() for p in first]
        opt.step()                                            # changes the weights through raw pointers, then repacks everything in one launch
        torch.cuda.synchronize()
        for (w, kw), p, b in zip(reqs, first, before):
            again = Fn.packed(w, **kw)
            assert again is p, "the cache entry was not left current: packed() launched its own pack"
            want = _fresh(w, **kw)
            assert torch.equal(p, want), (tuple(w.shape), kw)
            assert not torch.equal(p, b), "operand unchanged by a step that changed the weight"
            if hasattr(want, "_dmvae_kmajor"):
                assert torch.equal(p._dmvae_kmajor, want._dmvae_kmajor), (tuple(w.shape), kw, "K-tile-major copy")
    assert fp.pack_reg["n"] == len(reqs) and not fp.pack_reg["dirty"]


def test_registry_follows_replaced_and_new_operands(monkeypatch):
    """An operand packed for the first time after some steps joins the table; one whose cache entry was replaced behind the registry's back (external update
    of the flat buffer: the lazy route repacks) is re-registered; results stay those of the per-weight packs."""
    from dmvae_amd import functional as Fn
    from dmvae_amd.optim import FlatAdamWEMA, FlatParams
    monkeypatch.setattr(Fn, "_PACK_BATCHED", True)
    ps = _params()
    fp = FlatParams(ps, with_ema=False)
    opt = FlatAdamWEMA(fp, lr=1e-2, warmup_steps=0, max_norm=0.0)
    a = Fn.packed(ps[0])
    fp.grad.normal_()
    opt.step()
    b = Fn.packed(ps[2], for_dgrad=True)                      # new operand: table is rebuilt on the next step
    fp.flat.mul_(0.5)
    fp.after_external_update()                                # epoch bump without a repack: every cached operand is stale
    a2 = Fn.packed(ps[0])
    assert a2 is not a and torch.equal(a2, _fresh(ps[0]))
    fp.grad.normal_()
    opt.step()
    torch.cuda.synchronize()
    assert Fn.packed(ps[0]) is a2 and torch.equal(a2, _fresh(ps[0]))
    assert Fn.packed(ps[2], for_dgrad=True) is not None and torch.equal(Fn.packed(ps[2], for_dgrad=True), _fresh(ps[2], for_dgrad=True))
    assert fp.pack_reg["n"] == 2


def test_trainer_steps_are_bit_identical_with_and_without_batched_repack(monkeypatch):
    """Three TokenizerTrainer steps (reduced decoder): the batched repack changes when the operands are written, not what is written."""
    import warnings
    from dmvae_amd import functional as Fn
    from dmvae_amd.train import TokenizerTrainer
    from dmvae_amd.utils.lpips import LPIPS
    from test_oracle_golden import vae_tiny_params

    def run(batched):
        monkeypatch.setattr(Fn, "_PACK_BATCHED", batched)
        torch.manual_seed(0)
        p, vae = vae_tiny_params(seed=71, width=256)
        vae.load_state_dict(p, strict=True)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            lp = LPIPS().eval().requires_grad_(False)
        tr = TokenizerTrainer(vae.cuda(), lp.cuda(), lr=2e-6, warmup_steps=1)
        x = (torch.rand(2, 3, 256, 256, generator=torch.Generator().manual_seed(3)) * 2 - 1).cuda()
        logs = []
        for _ in range(3):
            tr.step(x)
            logs.append(tr.read_log())
        n = tr.fp.pack_reg["n"]
        return logs, tr.fp.flat.clone(), n
    l1, w1, n1 = run(True)
    l0, w0, n0 = run(False)
    assert n1 > 20 and n0 == 0
    assert l1 == l0 and torch.equal(w1, w0)
