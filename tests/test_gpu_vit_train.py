"""Trainable ViT encoder path on the HIP kernels (-m gpu): the backward-side kernels of csrc/vit_bwd.hip against fp64 autograd on the same
operands, and the block / encoder autograd Functions against the stock PyTorch module under autocast(bf16) and the CPU oracle."""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


@pytest.mark.parametrize("rows,c", [(37, 1024), (8224, 1024), (5, 256), (300, 1536)])
def test_layernorm_bwd(rows, c):
    from dmvae_amd import ops
    g = torch.Generator().manual_seed(rows + c)
    x = (torch.randn(rows, c, generator=g) * 2 + 0.3).to(DEV)
    gam, bet = (1 + 0.3 * torch.randn(c, generator=g)).to(DEV), torch.randn(c, generator=g).to(DEV)
    dy = torch.randn(rows, c, generator=g).to(DEV).to(BF)
    dres = torch.randn(rows, c, generator=g).to(DEV)
    xr = x.double().requires_grad_(True)
    gr, br = gam.double().requires_grad_(True), bet.double().requires_grad_(True)
    F.layer_norm(xr, (c,), gr, br, 1e-6).backward(dy.double())
    dx = dres.clone()
    dg, db = ops.layernorm_bwd_(dx, dy, x, gam, 1e-6)
    assert rel_err(dx, dres.double() + xr.grad) < 2e-6
    assert rel_err(dg, gr.grad) < 1e-5 and rel_err(db, br.grad) < 1e-5
    dx2 = dres.clone()
    dg2, db2 = ops.layernorm_bwd_(dx2, dy, x, gam, 1e-6)
    assert torch.equal(dx, dx2) and torch.equal(dg, dg2) and torch.equal(db, db2)        # fixed-order reductions
    acc_g, acc_b = dg.clone(), db.clone()
    ops.layernorm_bwd_(dres.clone(), dy, x, gam, 1e-6, dg_out=acc_g, db_out=acc_b, accumulate=True)
    assert rel_err(acc_g, 2 * gr.grad) < 1e-5 and rel_err(acc_b, 2 * br.grad) < 1e-5


@pytest.mark.parametrize("rows,c", [(37, 1024), (8224, 1024), (9, 256), (100, 512)])
def test_layerscale_bwd_and_gelu(rows, c):
    from dmvae_amd import ops
    g = torch.Generator().manual_seed(rows)
    dt = torch.randn(rows, c, generator=g).to(DEV)
    y = torch.randn(rows, c, generator=g).to(DEV).to(BF)
    gam = (0.5 + 0.2 * torch.randn(c, generator=g)).to(DEV)
    dy, dg = ops.layerscale_bwd(dt, y, gam)
    assert torch.equal(dy, (gam * dt).to(BF))
    assert rel_err(dg, (dt.double() * y.double()).sum(0)) < 1e-5
    x = (torch.randn(rows, c, generator=g) * 2).to(DEV).to(BF)
    d = torch.randn(rows, c, generator=g).to(DEV).to(BF)
    xr = x.double().requires_grad_(True)
    F.gelu(xr).backward(d.double())
    assert rel_err(ops.gelu(x).float(), F.gelu(xr.detach())) < 4e-3                       # bf16 result
    assert (ops.gelu(x).float() - F.gelu(x.float()).to(BF).float()).abs().max() <= 2 ** -7 * F.gelu(x.float()).abs().max()
    assert rel_err(ops.gelu_bwd(d, x).float(), xr.grad) < 4e-3


def _vit(embed_dim, depth, heads, img, seed=0):
    from dmvae_amd.models.vit import DinoV2ViT
    torch.manual_seed(seed)
    vit = DinoV2ViT(embed_dim=embed_dim, depth=depth, num_heads=heads, patch_size=16, img_size=img).to(DEV)
    with torch.no_grad():
        for blk in vit.blocks:                      # LayerScale at O(1) so that both branches matter; non-trivial norms / biases
            blk.ls1.gamma.uniform_(0.5, 1.5); blk.ls2.gamma.uniform_(0.5, 1.5)
        for n, p in vit.named_parameters():
            if n.endswith("bias"):
                p.normal_(0, 0.1)
            if "norm" in n and n.endswith("weight"):
                p.uniform_(0.7, 1.3)
            if n.endswith("fc1.weight") or n.endswith("fc2.weight") or n.endswith("qkv.weight") or n.endswith("proj.weight"):
                p.mul_(2.5)
        vit.cls_token.normal_(0, 0.5)
        vit.pos_embed.normal_(0, 0.5)
    return vit


def _rl2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def test_trainable_encoder_matches_stock_autocast():
    """forward_features with gradients on the HIP kernels vs the stock PyTorch modules under autocast(bf16) on the same weights: tokens,
    input gradient and every parameter gradient (rel-L2: bf16 activations on both sides, different rounding sites inside attention)."""
    import copy
    from dmvae_amd.models import vit_fast
    vit = _vit(256, 2, 4, 64)
    ref = copy.deepcopy(vit)
    assert vit_fast.hip_path_supported(vit, vit.pos_embed.shape[1])
    g = torch.Generator().manual_seed(1)
    x = torch.randn(3, 3, 64, 64, generator=g).to(DEV)
    dy = torch.randn(3, 17, 256, generator=g).to(DEV)
    xa = x.clone().requires_grad_(True)
    ya = vit.forward_features(xa)                                   # dispatches to the HIP path (trainable, CUDA, supported width)
    assert ya.dtype == BF and ya.shape == (3, 17, 256)
    (ya.float() * dy).sum().backward()
    xb = x.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=BF):
        yb = ref.forward_features_stock(xb)
        (yb.float() * dy).sum().backward()
    assert _rl2(ya.float(), yb.float()) < 1e-2
    assert _rl2(xa.grad, xb.grad) < 3e-2
    pa, pb = dict(vit.named_parameters()), dict(ref.named_parameters())
    for n in pa:
        assert pa[n].grad is not None, n
        assert _rl2(pa[n].grad, pb[n].grad) < 3e-2, (n, _rl2(pa[n].grad, pb[n].grad))
    # fp32 stock module (no autocast): the bf16 path stays within the bf16 noise floor of the exact gradients
    ref32 = copy.deepcopy(ref)
    ref32.zero_grad()
    xc = x.clone().requires_grad_(True)
    (ref32.forward_features_stock(xc) * dy).sum().backward()
    e_hip = max(_rl2(pa[n].grad, dict(ref32.named_parameters())[n].grad) for n in pa)
    e_stock = max(_rl2(pb[n].grad, dict(ref32.named_parameters())[n].grad) for n in pa)
    assert e_hip < 1.5 * e_stock + 5e-3, (e_hip, e_stock)


def test_trainable_encoder_full_width_is_deterministic():
    """ViT-L width (1024, 16 heads, 257 tokens), two blocks, batch 4: finite gradients everywhere, identical on a second run."""
    vit = _vit(1024, 2, 16, 256, seed=3)
    x = torch.randn(4, 3, 256, 256, generator=torch.Generator().manual_seed(2)).to(DEV)
    outs = []
    for _ in range(2):
        vit.zero_grad(set_to_none=True)
        xa = x.clone().requires_grad_(True)
        y = vit.forward_features(xa)
        assert y.shape == (4, 257, 1024)
        y.float().square().mean().backward()
        outs.append([y.detach().clone(), xa.grad.clone()] + [p.grad.clone() for p in vit.parameters()])
    for a, b in zip(*outs):
        assert torch.isfinite(a).all() and torch.equal(a, b)
    assert all(o.abs().max() > 0 for o in outs[0])


def test_vae_forward_with_trainable_encoder():
    """VAE.forward(freeze_encoder=False): gradients reach the encoder through the bottleneck MLP and the sliced token view."""
    import warnings
    from dmvae_amd.models.vae import VAE
    torch.manual_seed(5)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        vae = VAE(z_channels=32, model_size="base", encoder_kwargs=dict(embed_dim=256, depth=1, num_heads=4)).to(DEV)
    x = (torch.rand(2, 3, 256, 256, generator=torch.Generator().manual_seed(3)) * 2 - 1).to(DEV)
    rec = vae(x)
    assert rec.shape == (2, 3, 256, 256)
    (rec - x).abs().mean().backward()
    g = vae.encoder.model.blocks[0].attn.qkv.weight.grad
    assert g is not None and torch.isfinite(g).all() and g.abs().max() > 0
    assert vae.encoder.model.pos_embed.grad is not None and vae.encoder.model.patch_embed.proj.weight.grad is not None


def test_no_grad_encoder_reads_the_live_shadow_after_optimiser_steps():
    """An encoder whose parameters live in an optimiser's flat buffer with a bf16 shadow (DMDTrainer's VAE, train_dmd.py:518-524): between its training
    turns `vae.encode` runs it under no_grad through `frozen_forward_features`.  The fused AdamW step rewrites the shadow through raw pointers (neither
    data_ptr nor _version of a shadow view moves), so a Linear that cached a K-tile-major pack of that view would keep serving pre-step weights.  After
    every optimiser step the no-grad forward must equal the forward of a freshly built copy of the updated module, bit for bit."""
    import copy
    from dmvae_amd.models import vit_fast
    from dmvae_amd.optim import FlatAdamWEMA, FlatParams
    vit = _vit(512, 2, 8, 64, seed=7)                    # width 512: K >= 384, so the Linear layers take csrc/gemm_pp.hip (the route with the frozen pack)
    fp = FlatParams(list(vit.parameters()), with_ema=False)
    fp.enable_bf16_shadow()
    opt = FlatAdamWEMA(fp, lr=0.05, warmup_steps=0, max_norm=0.0, weight_decay=0.0)
    x = torch.randn(5, 3, 64, 64, generator=torch.Generator().manual_seed(3)).to(DEV)
    prev = None
    for step in range(3):
        with torch.no_grad():
            y = vit_fast.frozen_forward_features(vit, x).clone()
            fresh = copy.deepcopy(vit)                   # new parameter objects: no cached operand, no shadow -- converted from the current f32 weights
            for p in fresh.parameters():
                for a in ("_dmvae_shadow", "_dmvae_shadow_ver", "_dmvae_epoch", "_dmvae_pack_reg", "_dmvae_packed", "_dmvae_bf16", "_dmvae_grad_view"):
                    if hasattr(p, a):
                        delattr(p, a)
            y_ref = vit_fast.frozen_forward_features(fresh, x)
        assert torch.equal(y, y_ref), f"step {step}: stale weights in the no-grad encoder forward"
        if prev is not None:
            assert not torch.equal(y, prev), "the optimiser step did not change the output: the test would not see stale weights"
        prev = y
        fp.begin_step()
        fp.grad.copy_(torch.randn(fp.numel, generator=torch.Generator().manual_seed(10 + step)).to(DEV))
        opt.step()


def test_no_grad_trainable_forward_takes_the_fused_route_with_the_same_bits():
    """`trainable_forward_features` under no_grad (the DMD stage's student-only steps, train_dmd.py:520-523) runs the frozen route's fused launches (GELU in the
    fc1 epilogue, LayerScale + residual + next LayerNorm in one pass): the same tokens, bit for bit, as the block Functions' forward it replaces there."""
    from dmvae_amd.models import vit_fast
    vit = _vit(1024, 2, 16, 256, seed=11)
    x = torch.randn(3, 3, 256, 256, generator=torch.Generator().manual_seed(5)).to(DEV)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        vit_fast.NOGRAD_FUSED = False
        try:
            y_fn = vit_fast.trainable_forward_features(vit, x).clone()
        finally:
            vit_fast.NOGRAD_FUSED = True
        y_fused = vit_fast.trainable_forward_features(vit, x)
    assert y_fn.dtype == y_fused.dtype and torch.equal(y_fn, y_fused)
    with torch.autocast("cuda", dtype=torch.bfloat16):          # with gradients enabled nothing changes: the Function route, a graph
        y_g = vit_fast.trainable_forward_features(vit, x)
    assert y_g.requires_grad and torch.equal(y_g.detach(), y_fused)
