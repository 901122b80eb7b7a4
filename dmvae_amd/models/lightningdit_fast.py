"""Inference forward of LightningDiT on the HIP kernels (csrc/dit.hip) -- the route `LightningDiT.forward` takes when no gradient is needed
(the teacher's and the student's evaluations inside the DMD loss, train_dmd.py:211-217: four DiT-XL/1 forwards per VAE turn).

Same arithmetic as `forward_stock` under autocast(bf16), with bf16 rounding at the sites where the reference's autocast graph rounds (see
csrc/dit.hip); per block: 1 fused (gated residual +) RMSNorm+modulate, qkv GEMM, 1 fused QK-norm+RoPE+head split, attention in one fused kernel (csrc/vit.hip, head dims 64 and 72
alike: the staged head dim pads to 96; composed batched QK^T GEMM / f32 softmax / PV GEMM beyond 288 tokens), proj GEMM, 1 gated residual, RMSNorm+modulate, w12 GEMM, SwiGLU gate, w3 GEMM,
gated residual -- every token-level Linear on this build's GEMM kernel (`functional.linear`: csrc/gemm_pp.hip).
The per-sample pieces (timestep / label embedding, adaLN Linear: one row per sample) stay stock PyTorch under autocast."""
import os

import torch
import torch.nn.functional as F

from .. import ops

_BF = torch.bfloat16


def structurally_supported(model) -> bool:
    """The configuration the DiT kernels cover (what `supported` checks besides the call's tensor and autocast state).  On this route every op is per sample
    -- per token row, per (sample, head) -- so a 2B-sample call equals two B-sample calls bit for bit (train.DMDTrainer's batched cond / uncond evaluation)."""
    from .lightningdit import RMSNorm, SwiGLUFFN
    blk = model.blocks[0]
    c, hd = model.hidden_size, model.hidden_size // model.num_heads
    tokens = model.x_embedder.num_patches
    return bool(model.use_rope and model.use_rmsnorm and isinstance(blk.attn.q_norm, RMSNorm) and isinstance(blk.mlp, SwiGLUFFN) and not blk.wo_shift
                and c % 8 == 0 and c <= 2048 and hd % 2 == 0 and hd <= 128 and tokens % 32 == 0 and blk.mlp.w3.in_features % 8 == 0)


def supported(model, x: torch.Tensor) -> bool:
    if not (x.is_cuda and torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == _BF):
        return False
    return structurally_supported(model) and x.shape[-1] * x.shape[-2] == model.x_embedder.num_patches * model.patch_size ** 2


def _bf(p):
    """bf16 copy of a parameter, cached until it changes (functional._bf: in-place optimiser updates bump `_version`) -- the frozen teacher
    converts its 675 M parameters once, not on each of its two evaluations per step."""
    from ..functional import _bf as cached
    return cached(p)


BATCHED_ADALN = True      # forward_inference: every block's adaLN modulation Linear in one launch (False: one launch per block; tests compare)


def _parity_on() -> bool:
    from .. import parity
    return parity.on()


_FUSED_ATTN = True        # False: attention composed from batched GEMMs + softmax (tests compare)
_FUSED_QKNORM = True      # QK-norm + RoPE inside the attention kernel


def _attention(qkv, blk, rope, heads):
    b, n, c3 = qkv.shape
    c = c3 // 3
    d = c // heads
    if _FUSED_ATTN and _FUSED_QKNORM and ops.attention_heads_supported(n, d):
        return ops.attention_qknorm_rope(qkv, blk.attn.q_norm.weight, blk.attn.k_norm.weight, rope.freqs_cos, rope.freqs_sin, heads, blk.attn.q_norm.eps, d ** -0.5)
    q, k, v = ops.qknorm_rope(qkv, blk.attn.q_norm.weight, blk.attn.k_norm.weight, rope.freqs_cos, rope.freqs_sin, heads, blk.attn.q_norm.eps)
    if _FUSED_ATTN and ops.attention_heads_supported(n, d):
        return ops.attention_heads(q, k, v, b, d ** -0.5)
    p = ops.softmax_rows(ops.gemm_nt(q, k, out_f32=True), d ** -0.5)            # [B*H, N, N] bf16
    o = ops.gemm_nt(p, ops.transpose_last2(v))                                  # [B*H, N, D]
    return o.view(b, heads, n, d).permute(0, 2, 1, 3).reshape(b, n, c)


def _t_embed(model, t: torch.Tensor, train: bool = False) -> torch.Tensor:
    """TimestepEmbedder.forward (lightningdit.py:96-139) under autocast(bf16) on this build's kernels: sinusoidal features (f32, cast like autocast casts a
    Linear's input) -> Linear -> SiLU (on the bf16-rounded pre-activation, in the GEMM's epilogue when no gradient is needed) -> Linear; bf16 [B, C]."""
    from ..functional import LinearFn, SiluFn, linear
    te = model.t_embedder
    emb = te.timestep_embedding(t, te.frequency_embedding_size)
    l0, l2 = te.mlp[0], te.mlp[2]
    if train:
        return LinearFn.apply(SiluFn.apply(LinearFn.apply(emb, l0.weight, l0.bias)), l2.weight, l2.bias)
    return linear(linear(emb.to(_BF), l0.weight, l0.bias, ops.ACT_SILU), l2.weight, l2.bias)


@torch.no_grad()
def forward_inference(model, x: torch.Tensor, t: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """x [B,C,H,W], t [B], y [B] -> velocity [B,C_out,H,W] in bf16 (what the stock modules return under autocast)."""
    from ..functional import linear
    b, cin, hh, ww = x.shape
    ps, c, heads = model.patch_size, model.hidden_size, model.num_heads
    w = model.x_embedder.proj.weight
    patches = x.view(b, cin, hh // ps, ps, ww // ps, ps).permute(0, 2, 4, 1, 3, 5).reshape(b, -1, cin * ps * ps)
    h = linear(patches.to(_BF), _bf(w).view(w.shape[0], -1), _bf(model.x_embedder.proj.bias)).float() + model.pos_embed
    h = h.contiguous()
    n = h.shape[1]
    cvec = _t_embed(model, t) + model.y_embedder(y, False)                      # [B, C] f32: bf16 timestep embedding + the f32 label-embedding row
    sc = F.silu(cvec)
    # Every gated residual is folded into the RMSNorm that follows it (one pass over the residual stream instead of two), so a block's MLP residual is applied
    # by the NEXT block's norm1 -- or by the final layer's norm -- with that layer's modulation.
    scb = sc.to(_BF)
    fl = model.final_layer

    def adaln(lin):
        # [B, 6C]: shift_msa | scale_msa | gate_msa | shift_mlp | scale_mlp | gate_mlp.  One row per SAMPLE (16-64 rows against a 6912 x 1152 weight): a
        # GEMV-shaped, weight-bandwidth-bound problem -- `linear` sends it to the weight-streaming kernel (csrc/linear_rows.hip); more than 64 samples per call
        # go to the tile kernels
        return linear(scb, lin.weight, lin.bias)

    # the adaLN Linears of ALL blocks as one batched launch (csrc/linear_rows.hip, blockIdx.y = block): the same kernel and the same bits as one call per block
    lins = [blk.adaLN_modulation[1] for blk in model.blocks]
    mods = None
    if BATCHED_ADALN and lins and b <= 64 and ops.linear_rows_supported(b, lins[0].weight.shape[0], c) and not _parity_on():
        wb = [ww if ww.dtype == _BF else _bf(ww) for ww in (l.weight for l in lins)]
        bb = [bv if bv.dtype == _BF else _bf(bv) for bv in (l.bias for l in lins)]
        mods = ops.linear_rows_batched(scb, wb, bb)                               # [L, B, 6C]
    pend = None                                                                   # (y, mod) of the previous block's MLP branch, not yet added to h
    for i, blk in enumerate(model.blocks):
        mod = mods[i] if mods is not None else adaln(blk.adaLN_modulation[1])
        if pend is None:
            a = ops.rmsnorm_modulate(h, blk.norm1.weight, mod, 0, c, blk.norm1.eps)
        else:
            a = ops.gated_residual_rmsnorm_modulate_(h, pend[0], pend[1], 5 * c, blk.norm1.weight, mod, 0, c, blk.norm1.eps)
        qkv = linear(a, blk.attn.qkv.weight, blk.attn.qkv.bias)
        o = linear(_attention(qkv, blk, model.feat_rope, heads), blk.attn.proj.weight, blk.attn.proj.bias)
        a = ops.gated_residual_rmsnorm_modulate_(h, o, mod, 2 * c, blk.norm2.weight, mod, 3 * c, 4 * c, blk.norm2.eps)
        g = linear(a, blk.mlp.w12.weight, blk.mlp.w12.bias, ops.ACT_SWIGLU)        # silu(x1) * x2 in the GEMM's epilogue: x12 never reaches HBM
        pend = (linear(g, blk.mlp.w3.weight, blk.mlp.w3.bias), mod)
    mod = adaln(fl.adaLN_modulation[1])                                           # [B, 2C]: shift | scale
    if pend is None:
        a = ops.rmsnorm_modulate(h, fl.norm_final.weight, mod, 0, c, fl.norm_final.eps)
    else:
        a = ops.gated_residual_rmsnorm_modulate_(h, pend[0], pend[1], 5 * c, fl.norm_final.weight, mod, 0, c, fl.norm_final.eps)
    out = model.unpatchify(linear(a, fl.linear.weight, fl.linear.bias))
    if model.learn_sigma:
        out, _ = out.chunk(2, dim=1)
    return out


class GraphedInference:
    """`forward_inference` of a FROZEN model captured once in a hipGraph and replayed: `f(x, t, y)` copies the arguments into the graph's static inputs,
    replays ~330 kernel launches with one host call and returns the graph's static output buffer (overwritten by the next call -- consume it first).
    For loops that call the same model hundreds of times at one shape (the sampler: 250 steps per batch, where launching the kernels one by one
    leaves the GPU idle 10-15 % of the step).  Must be built and called under the autocast(bf16) context `forward_inference` needs.  The captured kernels
    read the cached bf16 copies of the weights: rebuild after the weights change."""

    def __init__(self, model, x: torch.Tensor, t: torch.Tensor, y: torch.Tensor, warmup: int = 2):
        if not supported(model, x):
            raise RuntimeError("GraphedInference needs the HIP inference route (CUDA tensors under torch.autocast('cuda', dtype=torch.bfloat16))")
        self.model = model
        self.x, self.t, self.y = x.detach().clone(), t.detach().clone(), y.detach().clone()
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side), torch.no_grad():                      # warm-up off the capture: lazy kernel attributes, GEMM heuristics, weight caches
            for _ in range(warmup):
                forward_inference(model, self.x, self.t, self.y)
        cur.wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        # thread-local capture mode: in the default (global) mode every OTHER thread's event query is illegal while the capture lasts -- RCCL's watchdog thread
        # polls its work events all the time (sample_50k.py runs under torchrun) and aborts the process with "operation not permitted when stream is capturing"
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"), torch.no_grad():
            self.out = forward_inference(model, self.x, self.t, self.y)

    def matches(self, x, t, y) -> bool:
        return x.shape == self.x.shape and x.dtype == self.x.dtype and t.shape == self.t.shape and y.shape == self.y.shape and x.device == self.x.device

    def __call__(self, x: torch.Tensor, t: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        assert self.matches(x, t, y), "GraphedInference: shapes differ from the captured ones"
        self.x.copy_(x)
        self.t.copy_(t)
        self.y.copy_(y)
        self.graph.replay()
        return self.out


STACK_SEGMENTS = 1        # autograd nodes the block stack is cut into in a single process
STACK_SEGMENTS_DP = 4     # ... and with more than one rank (gradient buckets become ready per segment: the all-reduce overlaps the rest of the backward pass)
STACK_FN = True      # forward_train: all blocks as one functional.DitStackFn node (False: one DitBlockFn per block + LinearFn modulations; tests compare the two)


def tokens1_supported(model, x: torch.Tensor) -> bool:
    """The one-token configuration of the 2-D toy (toy_example_2d/dmd.py:436-454: LightningDiT-Mini/1 with input_size = 1): RoPE + RMSNorm + SwiGLU blocks, any
    SwiGLU width (682 there), width a multiple of 8, under autocast(bf16) on the GPU."""
    from .lightningdit import RMSNorm, SwiGLUFFN
    if not (x.is_cuda and torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == _BF):
        return False
    blk = model.blocks[0]
    return bool(model.x_embedder.num_patches == 1 and x.shape[-1] * x.shape[-2] == model.patch_size ** 2 and model.use_rmsnorm and isinstance(blk.mlp, SwiGLUFFN)
                and isinstance(blk.norm1, RMSNorm) and not blk.wo_shift and model.hidden_size % 8 == 0 and model.hidden_size <= 2048)


def _swiglu_operands(blk):
    """w12 / b12 / w3 with the SwiGLU width H padded to a multiple of 32 -- rows [0, H) = x1's, [Hp, Hp + H) = x2's, the rest zero; w3's extra input columns zero --:
    LightningDiT-Mini's H = int(2/3 * 1024) = 682 is not a multiple of 8, the 16-byte granule of the bf16 kernels.  Padded columns give silu(0) * 0 = 0 and meet zero
    weights: the same function; torch.cat / pad are layout, their autograd slices the gradients back to the parameters' shapes."""
    w12, b12, w3 = blk.mlp.w12.weight, blk.mlp.w12.bias, blk.mlp.w3.weight
    h = w3.shape[1]
    hp = (h + 31) // 32 * 32
    if hp == h:
        return w12, b12, w3
    zw, zb = w12.new_zeros(hp - h, w12.shape[1]), b12.new_zeros(hp - h)
    return torch.cat([w12[:h], zw, w12[h:], zw]), torch.cat([b12[:h], zb, b12[h:], zb]), torch.nn.functional.pad(w3, (0, hp - h))


def forward_tokens1(model, x: torch.Tensor, t: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """LightningDiT.forward at ONE token per sample (config C1: every 2-D point is a [2, 1, 1] "image"), with or without gradients, on this build's kernels: with a
    single key the softmax is 1 and the attention output IS v -- q, k, their RMSNorm weights and RoPE do not reach the output and get exactly zero gradient, as in
    the reference --, so a block is five Linears (adaLN, qkv, proj, w12, w3) on the Linear GEMM kernels plus RMSNorm + modulate, the gated residuals and SwiGLU on
    csrc/dit.hip, composed from single autograd Functions (functional.LinearFn / RmsnormModulateFn / GatedResidualFn / SwigluFn).  The samples are the rows."""
    from ..functional import GatedResidualFn, LinearFn, RmsnormModulateFn, SwigluFn
    b, cin = x.shape[0], x.shape[1]
    ps, c = model.patch_size, model.hidden_size
    w = model.x_embedder.proj.weight
    train = torch.is_grad_enabled() and any(p.requires_grad for p in model.parameters())
    h = (LinearFn.apply(x.reshape(b, 1, cin * ps * ps), w.view(w.shape[0], -1), model.x_embedder.proj.bias).float() + model.pos_embed).contiguous()
    cvec = _t_embed(model, t, train=train).float() + model.y_embedder(y, model.training)
    sc = F.silu(cvec)
    for blk in model.blocks:
        lin = blk.adaLN_modulation[1]
        mod = LinearFn.apply(sc, lin.weight, lin.bias)                              # [B, 6C] bf16
        a1 = RmsnormModulateFn.apply(h, blk.norm1.weight, mod, 0, c, blk.norm1.eps)
        v = LinearFn.apply(a1, blk.attn.qkv.weight, blk.attn.qkv.bias)[..., 2 * c:]   # attention over one key = its value
        h = GatedResidualFn.apply(h, LinearFn.apply(v.contiguous(), blk.attn.proj.weight, blk.attn.proj.bias), mod, 2 * c)
        a2 = RmsnormModulateFn.apply(h, blk.norm2.weight, mod, 3 * c, 4 * c, blk.norm2.eps)
        w12, b12, w3 = _swiglu_operands(blk)
        g = SwigluFn.apply(LinearFn.apply(a2, w12, b12))
        h = GatedResidualFn.apply(h, LinearFn.apply(g, w3, blk.mlp.w3.bias), mod, 5 * c)
    fl = model.final_layer
    a = RmsnormModulateFn.apply(h, fl.norm_final.weight, LinearFn.apply(sc, fl.adaLN_modulation[1].weight, fl.adaLN_modulation[1].bias), 0, c, fl.norm_final.eps)
    out = model.unpatchify(LinearFn.apply(a, fl.linear.weight, fl.linear.bias))
    if model.learn_sigma:
        out, _ = out.chunk(2, dim=1)
    return out


def forward_train(model, x: torch.Tensor, t: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """`forward` with gradients (the student's flow-matching turn, train_dmd.py:565-575) under the caller's autocast(bf16): timestep / label embedders
    and the per-sample adaLN Linears through stock autograd, the patch embedding and the output Linear as `functional.LinearFn`, every block as one `functional.DitBlockFn`,
    the final norm as `RmsnormModulateFn`.  Label dropout as in the module (`y_embedder(y, model.training)`)."""
    from .. import functional as Fn
    from ..functional import DitBlockFn, DitStackFn, LinearFn, RmsnormModulateFn
    Fn._OWNED_GRADS.clear()
    b, cin, hh, ww = x.shape
    ps, c, heads = model.patch_size, model.hidden_size, model.num_heads
    w = model.x_embedder.proj.weight
    patches = x.view(b, cin, hh // ps, ps, ww // ps, ps).permute(0, 2, 4, 1, 3, 5).reshape(b, -1, cin * ps * ps)
    h = (LinearFn.apply(patches, w.view(w.shape[0], -1), model.x_embedder.proj.bias).float() + model.pos_embed).contiguous()
    cvec = _t_embed(model, t, train=True).float() + model.y_embedder(y, model.training)      # the embedding lookup (and its index-add backward) is not a GEMM
    sc = F.silu(cvec)                                                             # adaLN_modulation[0] of every block: the same f32 values, computed once
    rope = model.feat_rope
    stack = STACK_FN and Fn.dit_stack_supported(b, h.shape[1], c, heads)
    if stack:
        # One node per SEGMENT of blocks.  A single process takes all blocks as one segment; under data parallelism the gradients of a node reach autograd -- and
        # FlatGradSync's post-accumulate hooks, which start a bucket's all-reduce -- only when the node returns, so the stack is cut into STACK_SEGMENTS_DP nodes:
        # the collectives of the later blocks then run beside the backward pass of the earlier ones (2.7 GB of gradients for LightningDiT-XL/1, train_dmd.py:355)
        from .. import dist as _dist
        nseg = max(1, min(len(model.blocks), STACK_SEGMENTS_DP if _dist.get_world_size() > 1 else STACK_SEGMENTS))
        per = (len(model.blocks) + nseg - 1) // nseg
        for s0 in range(0, len(model.blocks), per):
            params = []
            for blk in model.blocks[s0:s0 + per]:
                params += [blk.norm1.weight, blk.attn.qkv.weight, blk.attn.qkv.bias, blk.attn.q_norm.weight, blk.attn.k_norm.weight, blk.attn.proj.weight,
                           blk.attn.proj.bias, blk.norm2.weight, blk.mlp.w12.weight, blk.mlp.w12.bias, blk.mlp.w3.weight, blk.mlp.w3.bias,
                           blk.adaLN_modulation[1].weight, blk.adaLN_modulation[1].bias]
            h = DitStackFn.apply(h, sc, rope.freqs_cos, rope.freqs_sin, heads, model.blocks[0].norm1.eps, *params)
    for blk in (() if stack else model.blocks):
        lin = blk.adaLN_modulation[1]
        mod = LinearFn.apply(sc, lin.weight, lin.bias)                            # [B, 6C] bf16; one row per sample: csrc/linear_rows.hip forward and input gradient
        h = DitBlockFn.apply(h, mod, blk.norm1.weight, blk.attn.qkv.weight, blk.attn.qkv.bias, blk.attn.q_norm.weight, blk.attn.k_norm.weight,
                             blk.attn.proj.weight, blk.attn.proj.bias, blk.norm2.weight, blk.mlp.w12.weight, blk.mlp.w12.bias, blk.mlp.w3.weight,
                             blk.mlp.w3.bias, rope.freqs_cos, rope.freqs_sin, heads, blk.norm1.eps)
    fl = model.final_layer
    a = RmsnormModulateFn.apply(h, fl.norm_final.weight, LinearFn.apply(sc, fl.adaLN_modulation[1].weight, fl.adaLN_modulation[1].bias), 0, c, fl.norm_final.eps)
    out = model.unpatchify(LinearFn.apply(a, fl.linear.weight, fl.linear.bias))
    if model.learn_sigma:
        out, _ = out.chunk(2, dim=1)
    return out
