/* dmvae_hip.h -- C ABI of libdmvae_hip.so, the MI355X (gfx950) kernels behind the DMVAE
 * training hot path.
 *
 * The reference (sen-ye/dmvae) is pure PyTorch and has no FFI of its own: the device work below
 * is what its nn.Module.forward()/autograd reaches through ATen (cuDNN conv, cuBLAS GEMM, native
 * GroupNorm, SDPA, elementwise loss ops).  Each entry point cites the reference call site whose
 * device work it replaces.  The binding a maintainer adds is a ctypes stub;
 * dmvae_amd/_lib.py is that stub.
 *
 * Conventions
 *  - all pointers are DEVICE pointers owned by the caller (PyTorch's caching allocator); kernels
 *    never allocate, free or synchronise; every call is enqueued on `stream`.
 *  - activations are NHWC ("channels last") bf16 unless stated; reductions / statistics are f32.
 *  - return 0 on success, negative errno-style code on failure (-22 bad argument, -5 launch
 *    failure); dmvae_last_error() returns a thread-local message.  Nothing throws or exits.
 *  - one process per GPU; calls may come from the autograd engine thread.
 */
#ifndef DMVAE_HIP_H
#define DMVAE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* dmvae_stream_t; /* == hipStream_t */

const char* dmvae_last_error(void);
/* ABI version; bumped when a signature changes. */
int dmvae_abi_version(void);

/* ---- convolution / GEMM (MFMA-bound) -------------------------------------------------------- */

typedef struct dmvae_conv_desc {
  int32_t n, h, w;   /* input batch, height, width (pre-upsample) */
  int32_t cin, cout;
  int32_t ks;        /* 1 or 3 (3 => padding 1, stride 1) */
  int32_t upsample;  /* 1: nearest x2 of the input folded into the gather (flux_ae.py:103-107) */
  int32_t act;       /* epilogue: 0 none, 1 SiLU (vae.py:60), 2 ReLU (lpips.py VGG trunk) */
  int32_t out_f32;   /* 1: y is float32 (parity / final layers), else bf16 */
} dmvae_conv_desc;

/* y[n,ho,wo,cout] = act( conv(x, w) + bias + residual ).
 * x: [n,h,w,cin] bf16; w: [cout, ks*ks, cin] bf16 (tap-major, see dmvae_pack_conv_weight);
 * bias: [cout] f32 or NULL; residual: [n,ho,wo,cout] bf16 or NULL.
 * Replaces nn.Conv2d at models/flux_ae.py:32-35,63,65,67,101,237,274 and nn.Linear at
 * models/vae.py:58-62 (ks=1,h=w=1,n=tokens).  dgrad of a stride-1 conv is this same call with
 * the weights packed by dmvae_pack_conv_weight(..., for_dgrad=1). */
int dmvae_conv2d_nhwc_fwd(const void* x, const void* w, const void* bias, const void* residual, void* y,
                          const dmvae_conv_desc* d, dmvae_stream_t stream);

/* Weight/bias gradient of the conv above (reduction over pixels, split-K, deterministic).
 * dy: [n,ho,wo,cout] bf16; a: the conv's bf16 input [n,h,w,cin] (pre-upsample when d->upsample);
 * dw: [cout][cin][ks][ks] f32 (PyTorch nn.Conv2d.weight layout); dbias: [cout] f32 or NULL.
 * accumulate=1 adds into dw/dbias (autograd .grad accumulation), 0 overwrites.
 * workspace: >= dmvae_conv2d_nhwc_wgrad_workspace(d) bytes, caller-owned.  d->act/out_f32 ignored.
 * Replaces autograd's conv/linear weight-gradient for the call sites listed at conv2d_nhwc_fwd. */
size_t dmvae_conv2d_nhwc_wgrad_workspace(const dmvae_conv_desc* d);
int dmvae_conv2d_nhwc_wgrad(const void* dy, const void* a, void* dw, void* dbias, void* workspace,
                            size_t workspace_bytes, const dmvae_conv_desc* d, int accumulate,
                            dmvae_stream_t stream);

/* ---- GroupNorm(+swish) on NHWC bf16 (HBM-bound) --------------------------------------------- */

/* Workspace bytes needed by groupnorm_stats / groupnorm_bwd for x: [n, hw, c]; 0 if unsupported
 * (c must be a multiple of 8 and <= 512, c % groups == 0). */
size_t dmvae_groupnorm_workspace(int n, int hw, int c, int groups);

/* stats[n][groups][2] = (mean, rstd) over (hw, c/groups) of x [n,hw,c] bf16, f32 accumulation.
 * Replaces the statistics half of nn.GroupNorm(32, C, eps=1e-6) (models/flux_ae.py:28,62,64,236). */
int dmvae_groupnorm_stats(const void* x, void* stats, void* workspace, size_t workspace_bytes, int n, int hw,
                          int c, int groups, float eps, dmvae_stream_t stream);

/* y = act((x-mean)*rstd*gamma+beta) as bf16; act = swish (x*sigmoid(x), flux_ae.py:21-22) when
 * swish!=0, identity otherwise (AttnBlock.norm, flux_ae.py:38). gamma/beta: [c] f32. */
int dmvae_groupnorm_apply(const void* x, const void* stats, const void* gamma, const void* beta, void* y,
                          int n, int hw, int c, int groups, int swish, dmvae_stream_t stream);

/* Backward of y=act(GN(x)): given da=dL/dy (bf16), x, stats, gamma, beta computes
 * dx (bf16) = GN/swish backward [+ dres when dres != NULL, fusing the residual-branch add],
 * dgamma/dbeta ([c] f32, accumulate!=0 adds) -- both may be NULL to skip. */
int dmvae_groupnorm_bwd(const void* da, const void* x, const void* dres, const void* stats, const void* gamma,
                        const void* beta, void* dx, void* dgamma, void* dbeta, void* workspace,
                        size_t workspace_bytes, int n, int hw, int c, int groups, int swish, int accumulate,
                        dmvae_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DMVAE_HIP_H */
