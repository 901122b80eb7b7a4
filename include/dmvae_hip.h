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
/* ABI version; bumped when a signature or a struct changes.  2: dmvae_conv_desc gained its last field, w_layout (zero = the behaviour of version 1).
 * 3: struct dmvae_pack_entry and the entry points that take it (dmvae_pack_weights_batched), dmvae_linear_bf16 / _plan / dmvae_linear_weight_t_kmajor.
 * 7: dmvae_linear_wgrad_grouped_plan / _xcd / _chunk_bytes, dmvae_conv_k4c1_*.
 * 4: dmvae_norm_conv_out_bwd / _supported / _workspace, dmvae_norm_conv_out_fwd / _supported, dmvae_conv_in3 / _supported / _workspace, dmvae_im2col_nhwc_taps, dmvae_linear_bf16_batched / _supported, dmvae_conv_to_image / _supported.
 * 5: dmvae_groupnorm_apply_short / _bwd_short / _short_supported / _bwd_short_workspace, dmvae_lpips_diff_pool.
 * 6: the whole-stack LightningDiT backward (dmvae_dit_boundary_bwd, dmvae_dit_stack_*, dmvae_colsum2_batched, dmvae_qknorm_rope_bwd_partial / _nblk), the batched
 *    per-sample Linears (dmvae_linear_rows_batched_bf16, dmvae_linear_rows_wgrad_batched), dmvae_linear_weight_t_kmajor_batched / dmvae_wt_entry_bytes, the grouped
 *    Linear weight gradients (dmvae_linear_wgrad_grouped*). */
int dmvae_abi_version(void);

/* ---- convolution / GEMM (MFMA-bound) -------------------------------------------------------- */

typedef struct dmvae_conv_desc {
  int32_t n, h, w;   /* input batch, height, width (pre-upsample) */
  int32_t cin, cout;
  int32_t ks;        /* 1, 3 (padding 1, stride 1 unless `stride` says otherwise) or 4 (padding 1, stride 1 or 2: the PatchGAN
                        convs, models/patchgan.py:125-147; y is [n, (h-2)/stride+1, (w-2)/stride+1, cout]) */
  int32_t upsample;  /* 1: nearest x2 of the input folded into the gather (flux_ae.py:103-107);
                        2: zero-insertion x2 (input pixel (y,x) sits at output position (2y+1,2x+1), zeros elsewhere): with
                           for_dgrad-packed weights this is the input gradient of the stride-2 Downsample conv below */
  int32_t act;       /* epilogue: 0 none, 1 SiLU (vae.py:60), 2 ReLU (lpips.py VGG trunk), 3 ReLU-backward mask: `residual` is
                        not added but gates the result, y = residual > 0 ? conv + bias : 0 (input gradient through conv+ReLU);
                        4 LeakyReLU(0.2) (models/patchgan.py:125) */
  int32_t out_f32;   /* 1: y is float32 (parity / final layers), else bf16 */
  int32_t stride;    /* 0 or 1: stride 1.  2: the Downsample conv of flux_ae.py:85-95 -- input zero-padded by one row/column at the
                        bottom/right only, 3x3, stride 2, no other padding: y is [n, h/2, w/2, cout] (h, w even; ks=3, upsample=0) */
  int32_t transposed;/* 1 (ks 4, or ks 3 with stride 2): x is the OUTPUT gradient [n,h,w,cin] of the conv described by (ks, stride) and y its
                        input gradient [n, (h-1)*stride+2, ..., cout] (ks 4) / [n, 2h, 2w, cout] (ks 3); weights packed with for_dgrad=1.
                        Forward entry point only. */
  int32_t w_layout;  /* forward entry points: 0 = w is [cout][ks*ks][cin] (tap-major); 1 = K-tile-major [cin/32][ks*ks][cout][32], the second operand
                        dmvae_pack_conv_weight_v2 writes -- a (32-channel chunk, tap) tile of the weights is then one contiguous run of whole 128-B lines
                        (in the tap-major layout it is `cout` half-lines 2*ks*ks*cin bytes apart).  Accepted only where dmvae_conv_kmajor_applies(d) is 1
                        (the large-tile conv kernel); any other shape with w_layout = 1 is an error, not a fallback. */
} dmvae_conv_desc;

/* 1 when conv2d_nhwc_fwd / _gnstats run descriptor d on the kx-halo kernel (plain 3x3, bf16 result, large shapes: csrc/conv_pp.hip) and therefore accept
 * w_layout = 1; 0 otherwise.  Pure function of d.  Reference: Helper of dmvae_conv2d_nhwc_fwd (every nn.Conv2d of models/flux_ae.py:21-107,239-269 and of the LPIPS trunk utils/lpips.py:116-153). */
int dmvae_conv_halo_applies(const dmvae_conv_desc* d);
/* 1 when the call runs on the large-tile conv kernel at all (csrc/conv_pp.hip, any instantiation: plain 3x3, 1x1, the 4x4 stride-2 conv and its per-parity
 * transpose, Upsample's folded gather) and may therefore carry w_layout = 1; dmvae_conv_halo_applies(d) = 1 is the subset that runs the kx-halo form.  Reference: Helper of dmvae_conv2d_nhwc_fwd (every nn.Conv2d of models/flux_ae.py:21-107,239-269 and of the LPIPS trunk utils/lpips.py:116-153). */
int dmvae_conv_kmajor_applies(const dmvae_conv_desc* d);

/* y[n,ho,wo,cout] = act( conv(x, w) + bias + residual ).
 * x: [n,h,w,cin] bf16; w: [cout, ks*ks, cin] bf16 (tap-major, see dmvae_pack_conv_weight; or its K-tile-major copy with d->w_layout = 1);
 * bias: [cout] f32 or NULL; residual: [n,ho,wo,cout] bf16 or NULL.
 * Replaces nn.Conv2d at models/flux_ae.py:32-35,63,65,67,101,237,274 and nn.Linear at
 * models/vae.py:58-62 (ks=1,h=w=1,n=tokens).  dgrad of a stride-1 conv is this same call with
 * the weights packed by dmvae_pack_conv_weight(..., for_dgrad=1). */
int dmvae_conv2d_nhwc_fwd(const void* x, const void* w, const void* bias, const void* residual, void* y,
                          const dmvae_conv_desc* d, dmvae_stream_t stream);

/* The conv above with the GroupNorm statistics of its (bf16) result as a by-product: stats[n][groups][2] = (mean, rstd) over (ho*wo, cout/groups) of y,
 * exactly what dmvae_groupnorm_stats(y) returns up to f32 summation order.  Most convs of flux_ae.Decoder feed a GroupNorm (models/flux_ae.py:62,64,71-76,
 * 236,266): for the large shapes the conv kernel sums its rounded results per tile in the epilogue and a finishing kernel combines them, so the
 * statistics pass -- a full read of y -- disappears; other shapes run the conv, then the statistics pass.  out_f32 must be 0.
 * workspace >= dmvae_conv2d_nhwc_fwd_gnstats_workspace(d, groups) bytes. */
size_t dmvae_conv2d_nhwc_fwd_gnstats_workspace(const dmvae_conv_desc* d, int groups);
int dmvae_conv2d_nhwc_fwd_gnstats(const void* x, const void* w, const void* bias, const void* residual, void* y, void* stats, void* workspace,
                                  size_t workspace_bytes, int groups, float eps, const dmvae_conv_desc* d, dmvae_stream_t stream);

/* Weight/bias gradient of the conv above (reduction over pixels, split-K, deterministic).
 * dy: [n,ho,wo,cout] bf16; a: the conv's bf16 input [n,h,w,cin] (pre-upsample when d->upsample);
 * dw: [cout][cin][ks][ks] f32 (PyTorch nn.Conv2d.weight layout); dbias: [cout] f32 or NULL.
 * accumulate=1 adds into dw/dbias (autograd .grad accumulation), 0 overwrites.
 * workspace: >= dmvae_conv2d_nhwc_wgrad_workspace(d) bytes, caller-owned.  d->act/out_f32 ignored.
 * Replaces autograd's conv/linear weight-gradient for the call sites listed at conv2d_nhwc_fwd.  Reference: Replaces autograd's weight / bias gradient of every nn.Conv2d of models/flux_ae.py:21-107,239-269 and of the LPIPS trunk utils/lpips.py:116-153 (loss.backward(), train_tokenizer.py:382). */
size_t dmvae_conv2d_nhwc_wgrad_workspace(const dmvae_conv_desc* d);
int dmvae_conv2d_nhwc_wgrad(const void* dy, const void* a, void* dw, void* dbias, void* workspace,
                            size_t workspace_bytes, const dmvae_conv_desc* d, int accumulate,
                            dmvae_stream_t stream);

/* Weight gradient of a 3x3 stride-1 conv with <= 4 output channels and Cin = 128 (the decoder's conv_out, models/flux_ae.py:237,274) straight from the
 * image gradient as autograd hands it over: dy [n][cout][h][w] f32 (NCHW), a [n][h][w][128] bf16 (the conv's input), dw [cout][128][3][3] f32
 * ((+)= with accumulate).  `a` is read once (csrc/wgrad_thin.hip); w % 32 == 0.  workspace >= dmvae_conv_out_wgrad_workspace(...) bytes (0: unsupported shape).
 * Replaces autograd's conv weight gradient at that site (the bias gradient is the plain sum of dy over n, h, w). */
size_t dmvae_conv_out_wgrad_workspace(int n, int h, int w, int cin, int cout);
int dmvae_conv_out_wgrad(const void* dy, const void* a, void* dw, void* workspace, size_t workspace_bytes, int n, int h, int w, int cin, int cout,
                         int accumulate, dmvae_stream_t stream);

/* ---- GroupNorm(+swish) on NHWC bf16 (HBM-bound) --------------------------------------------- */

/* Workspace bytes needed by groupnorm_stats / groupnorm_bwd for x: [n, hw, c]; 0 if unsupported
 * (c must be a multiple of 8 and <= 512, c % groups == 0).  Reference: Normalize() = GroupNorm(32, eps 1e-6), models/flux_ae.py:28, at :62,64,236. */
size_t dmvae_groupnorm_workspace(int n, int hw, int c, int groups);

/* stats[n][groups][2] = (mean, rstd) over (hw, c/groups) of x [n,hw,c] bf16, f32 accumulation.
 * Replaces the statistics half of nn.GroupNorm(32, C, eps=1e-6) (models/flux_ae.py:28,62,64,236). */
int dmvae_groupnorm_stats(const void* x, void* stats, void* workspace, size_t workspace_bytes, int n, int hw,
                          int c, int groups, float eps, dmvae_stream_t stream);

/* y = act((x-mean)*rstd*gamma+beta) as bf16; act: 0 identity (AttnBlock.norm, flux_ae.py:38), 1 swish (x*sigmoid(x),
 * flux_ae.py:21-22), 2 LeakyReLU(0.2).  gamma/beta: [c] f32.
 * With n=1, hw=N*H*W, groups=c this is nn.BatchNorm2d / nn.SyncBatchNorm (+LeakyReLU) of models/patchgan.py:134-145 on an
 * NHWC tensor: `stats` then holds the per-channel (mean, rstd) -- batch statistics in training (combined over ranks by the
 * caller for SyncBatchNorm), (running_mean, 1/sqrt(running_var+eps)) in eval. */
int dmvae_groupnorm_apply(const void* x, const void* stats, const void* gamma, const void* beta, void* y,
                          int n, int hw, int c, int groups, int act, dmvae_stream_t stream);

/* Backward of y=act(GN(x)): given da=dL/dy (bf16), x, stats, gamma, beta computes
 * dx (bf16) = GN/act backward [+ dres when dres != NULL, fusing the residual-branch add],
 * dgamma/dbeta ([c] f32, accumulate!=0 adds) -- both may be NULL to skip.  = bwd_reduce followed by bwd_apply.  Reference: Autograd of swish(Normalize(x)), models/flux_ae.py:24,28,62-65,236. */
int dmvae_groupnorm_bwd(const void* da, const void* x, const void* dres, const void* stats, const void* gamma,
                        const void* beta, void* dx, void* dgamma, void* dbeta, void* workspace,
                        size_t workspace_bytes, int n, int hw, int c, int groups, int act, int accumulate,
                        dmvae_stream_t stream);
/* dmvae_groupnorm_bwd with one more by-product: colsum[c] (+)= sum over (n, hw) of the dx it stores (bf16-rounded, f32 sum, fixed order) -- when dx is the
 * output gradient of a conv (a ResnetBlock's norm1 behind Upsample's conv, flux_ae.py:71,103-107), that is the conv's bias gradient, for free. */
int dmvae_groupnorm_bwd_colsum(const void* da, const void* x, const void* dres, const void* stats, const void* gamma, const void* beta, void* dx,
                               void* dgamma, void* dbeta, void* colsum, void* workspace, size_t workspace_bytes, int n, int hw, int c, int groups, int act,
                               int accumulate, int colsum_accumulate, dmvae_stream_t stream);
/* The two halves, for callers that own the statistics (SyncBatchNorm: all-reduce `sums` over ranks in between and pass
 * inv_count = 1 / global element count; eval-mode BatchNorm: skip the reduce and pass zero sums).
 * sums: [n][groups][2] f32 = (sum g, sum g*x_hat), g = da*act'(.)*gamma.  inv_count <= 0 selects 1/(hw*c/groups).  Reference: nn.BatchNorm2d / SyncBatchNorm backward of the discriminator, models/patchgan.py:133,141 (train_tokenizer.py:283-285). */
int dmvae_groupnorm_bwd_reduce(const void* da, const void* x, const void* stats, const void* gamma, const void* beta,
                               void* sums, void* dgamma, void* dbeta, void* workspace, size_t workspace_bytes, int n,
                               int hw, int c, int groups, int act, int accumulate, dmvae_stream_t stream);
int dmvae_groupnorm_bwd_apply(const void* da, const void* x, const void* dres, const void* stats, const void* sums,
                              const void* gamma, const void* beta, void* dx, int n, int hw, int c, int groups, int act,
                              float inv_count, dmvae_stream_t stream);

/* Backward of the decoder's tail, conv_out(swish(norm_out(x))) (flux_ae.py:266-268), with respect to x / gamma / beta in two passes over x: the
 * input-gradient conv (3 -> c channels, 27 multiply-adds per element) is evaluated on the matrix cores INSIDE both GroupNorm backward passes instead of being
 * stored by one kernel and read by two (csrc/groupnorm.hip::convout_bwd_kernel).  dy: the image gradient f32 [n][3][h][w] (NCHW, as autograd hands it over;
 * rounded to bf16 like the stored-operand route's), w: conv_out.weight f32 [3][c][3][3], x: norm_out's input bf16 [n][h][w][c], stats from
 * dmvae_groupnorm_stats; dx bf16 like x; dgamma / dbeta [c] f32 (both or neither; accumulate != 0 adds).  Shapes: c = 128, w % 16 == 0, cout = 3
 * (dmvae_norm_conv_out_bwd_supported); equals dmvae_conv2d_nhwc_fwd (the input-gradient form) followed by dmvae_groupnorm_bwd up to the summation order inside
 * one bf16 rounding of the intermediate. */
/* Forward of the same tail in one launch (csrc/conv_thin.hip, NORM instantiation): a = swish(GroupNorm(x)) is computed on the way into the conv's halo tile and
 * written out once (bf16 [n][h][w][c]: what the weight gradient of conv_out reads in the backward), y = conv_out(a) + bias leaves as the NCHW f32 image
 * [n][cout][h][w].  w: conv_out's packed bf16 operand [4][9][c] (dmvae_pack_conv_weight, rows_pad 4), bias f32 [cout] or NULL.  Shapes: c = 128, h % 4 == 0,
 * w % 32 == 0, cout <= 4.  Same bits as dmvae_groupnorm_apply -> dmvae_conv2d_nhwc_fwd (out_f32) -> dmvae_nhwc_to_nchw_f32.  Reference: models/flux_ae.py:266-268. */
int dmvae_norm_conv_out_fwd_supported(int n, int h, int w, int c, int groups, int cout);
int dmvae_norm_conv_out_fwd(const void* x, const void* stats, const void* gamma, const void* beta, const void* w, const void* bias, void* a, void* y,
                            int n, int h, int wd, int c, int groups, int cout, dmvae_stream_t stream);
/* 3x3 stride-1 conv of an NHWC bf16 tensor (cin = 64 / 128) to cout <= 4 channels with the result as an NCHW f32 image [n][cout][h][w], optionally multiplied per
 * output channel by mul[cout] (device) last: the LPIPS trunk's image gradient (utils/lpips.py:81-104 backward: VGG conv1_1's input gradient, / ScalingLayer.scale,
 * x the incoming gradient) in one launch.  w: the packed bf16 operand [4][9][cin] (dmvae_pack_conv_weight, rows_pad 4); bias f32 [cout] or NULL. */
int dmvae_conv_to_image_supported(int n, int h, int w, int cin, int cout);
int dmvae_conv_to_image(const void* x, const void* w, const void* bias, const void* mul, void* y, int n, int h, int wd, int cin, int cout, dmvae_stream_t stream);
int dmvae_norm_conv_out_bwd_supported(int n, int h, int w, int c, int groups, int cout);
size_t dmvae_norm_conv_out_bwd_workspace(int n, int h, int w, int c, int groups);
int dmvae_norm_conv_out_bwd(const void* dy, const void* w, const void* x, const void* stats, const void* gamma, const void* beta, void* dx, void* dgamma,
                            void* dbeta, void* workspace, size_t workspace_bytes, int n, int h, int wd, int c, int groups, int cout, int accumulate,
                            dmvae_stream_t stream);

/* A ResnetBlock's 1x1 nin_shortcut (models/flux_ae.py:67,77-82) evaluated on the matrix cores inside the block's first GroupNorm passes (csrc/norm_short.hip) instead
 * of as launches of its own over tensors those passes stream anyway.  Shapes: c = 256 channels of the block input, cs = 128 of the block output, hw % 16 == 0
 * (dmvae_groupnorm_short_supported).
 *   dmvae_groupnorm_apply_short: a = act(GroupNorm(x)) (bf16 [n][hw][c], dmvae_groupnorm_apply's arithmetic) and xs = bf16(x W^T + bias) (bf16 [n][hw][cs]) from one
 *     read of x; w: the shortcut's packed forward operand bf16 [cs][c] (dmvae_pack_conv_weight, for_dgrad = 0), bias f32 [cs] or NULL.  xs equals
 *     dmvae_conv2d_nhwc_fwd (ks 1) up to the f32 summation order inside its bf16 rounding.
 *   dmvae_groupnorm_bwd_short: dmvae_groupnorm_bwd (dmvae_groupnorm_bwd_colsum when colsum != NULL) with dres = bf16(dys W) computed in place: dys is the block's
 *     output gradient bf16 [n][hw][cs], wt the shortcut's packed input-gradient operand bf16 [c][cs] (for_dgrad = 1).  Equals the ks-1 input-gradient conv followed by
 *     dmvae_groupnorm_bwd(_colsum) up to the summation order inside the bf16 rounding of the shortcut gradient (colsum: and the order of its f32 partial sums). */
int dmvae_groupnorm_short_supported(int n, int hw, int c, int cs, int groups);
int dmvae_groupnorm_apply_short(const void* x, const void* stats, const void* gamma, const void* beta, const void* w, const void* bias, void* a, void* xs,
                                int n, int hw, int c, int cs, int groups, int act, dmvae_stream_t stream);
size_t dmvae_groupnorm_bwd_short_workspace(int n, int hw, int c, int cs, int groups);
int dmvae_groupnorm_bwd_short(const void* da, const void* x, const void* dys, const void* wt, const void* stats, const void* gamma, const void* beta,
                              void* dx, void* dgamma, void* dbeta, void* colsum, void* workspace, size_t workspace_bytes, int n, int hw, int c, int cs,
                              int groups, int act, int accumulate, int colsum_accumulate, dmvae_stream_t stream);

/* 3x3 stride-1 conv FROM THREE input channels (an NCHW f32 image) to cout = 64 / 128 channels with bias and ReLU, NHWC bf16 result: the first layer of the
 * LPIPS trunk behind its ScalingLayer (utils/lpips.py:81-104,116-135: VGG16 conv1_1 on both branches).  The n images come from one or two tensors (x0: the first
 * n0 images, x1: the rest; x1 may be NULL when n0 == n) -- no concatenated copy; shift / scale: device pointers to the ScalingLayer's three values each, applied
 * as (x - shift) / scale in f32 before the bf16 rounding of the operand, or both NULL.  w: f32 [cout][3][3][3] (the nn.Conv2d parameter), bias f32 [cout] or
 * NULL, act 0 (none) / 2 (ReLU).  Equals the zero-padded 32-channel route (nchw_to_nhwc + dmvae_conv2d_nhwc_fwd) up to the f32 summation order inside the
 * bf16 rounding of the result.  Shapes: w % 16 == 0 (dmvae_conv_in3_supported); workspace: the zero-bordered 4-channel bf16 copy of the images. */
int dmvae_conv_in3_supported(int n, int h, int w, int cout);
size_t dmvae_conv_in3_workspace(int n, int h, int w);
int dmvae_conv_in3(const void* x0, const void* x1, int n0, const void* shift, const void* scale, const void* w, const void* bias, void* y,
                   void* workspace, size_t workspace_bytes, int n, int h, int wd, int cout, int act, dmvae_stream_t stream);

/* ---- batched GEMMs on the same MFMA cores (decoder self-attention, flux_ae.py:37-49) --------- */

/* C[b][m][n] = act( sum_k A[b][m][k]*B[b][n][k] + bias[n] + R[b][m][n] ); row-major bf16, both
 * operands K-contiguous; *_bs are element strides between batch items (0 = shared operand);
 * C/R share c_bs.  K%32==0, N%4==0.  act as in dmvae_conv_desc; out_f32 selects C's type.  Reference: AttnBlock's q k^T / p v and the 1x1 convs around them, models/flux_ae.py:37-49; timm Mlp's fc1 / fc2 through models/vae.py:47-53. */
int dmvae_gemm_nt_batched(const void* A, const void* B, const void* bias, const void* R, void* C, int M, int N,
                          int K, int batch, long long a_bs, long long b_bs, long long c_bs, int act, int out_f32,
                          dmvae_stream_t stream);

/* C[b][m][n] = alpha * sum_k A[b][k][m]*B[b][k][n]  (A:[K][M], B:[K][N] row-major bf16; the
 * reduction dim is the slow one -> LDS transpose reads).  M%8==0, N%8==0.  Split-K, deterministic.  Reference: The weight-gradient products of the same layers (autograd of models/flux_ae.py:37-49, models/vae.py:47-53). */
size_t dmvae_gemm_tn_batched_workspace(int M, int N, int K, int batch);
int dmvae_gemm_tn_batched(const void* A, const void* B, void* C, void* workspace, size_t workspace_bytes, int M,
                          int N, int K, int batch, long long a_bs, long long b_bs, long long c_bs, float alpha,
                          int out_f32, dmvae_stream_t stream);

/* nn.Linear on a handful of rows -- one per SAMPLE: Y[M][N] = act(X[M][K] . W[N][K]^T + bias[N]), 1 <= M <= 64, K % 32 == 0, N % 4 == 0; x / w row-major
 * bf16 with leading dimensions ldx / ldw (multiples of 8), y [M][ldy] bf16 (f32 when out_f32).  Replaces the library call behind the per-sample conditioning
 * Linears of LightningDiT: adaLN_modulation[1] of every block and of the final layer (diffusion/lightningdit/lightningdit.py:236-240,266-268), the two Linears
 * of TimestepEmbedder.mlp (:96-139), and -- with w := the transposed bf16 copy [K_in][N_out] of the weight -- their input gradients.  Weight-bandwidth bound
 * (the 6912 x 1152 adaLN weight is read once per call); csrc/linear_rows.hip.  bias f32 [N] (bf16 when bias_bf16) or NULL; act 0 none, 1 SiLU on the bf16-rounded
 * pre-activation (bit-identical to act 0 + dmvae_silu_fwd; bf16 result only).  Deterministic (fixed-order in-block reduction). */
int dmvae_linear_rows_supported(int M, int N, int K);
int dmvae_linear_rows_bf16(const void* x, const void* w, const void* bias, void* y, int M, int N, int K, int ldx, int ldw, int ldy, int act,
                           int bias_bf16, int out_f32, int w_layout, dmvae_stream_t stream);
/* w_layout = 1: w is the K-tile-major copy [K / 32][N][32] (dmvae_linear_weight_t_kmajor writes it from the bf16 weight in ~8 us: what the input gradient
 * reads instead of a row-major transposed copy re-packed from the f32 master every step); ldw is then ignored.
 * Weight and bias gradient of the same per-sample Linear: dW[N][K] f32 (+)= dY[M][N]^T . X[M][K], db[N] f32 (+)= column sums of dY (db may be NULL);
 * 1 <= M <= 64, K % 8 == 0; dy / x bf16 row-major with leading dimensions lddy / ldx.  Bound by writing the f32 gradient; summed over the samples in order.  Reference: Autograd of adaLN_modulation[1] / the embedders' Linears, diffusion/lightningdit/lightningdit.py:236-250,262-264. */
int dmvae_linear_rows_wgrad(const void* dy, const void* x, void* dw, void* db, int M, int N, int K, int lddy, int ldx, int accumulate, dmvae_stream_t stream);

/* Dynamic tile claiming in the large-shape conv kernel (blocks claim tiles from per-XCD counters instead of a static stride, so that a launch next to another
 * stream's resident kernel -- an RCCL all-reduce overlapped with backward, utils/dist.py / train_tokenizer.py:302 -- slows down by the fraction of CUs taken
 * instead of a whole block-time): on = 1 / 0 for every launch from now on, -1 = the DMVAE_PP_DYNAMIC environment default.  Returns the value in force.
 * Results do not depend on it. */
int dmvae_set_dynamic(int on);

/* nn.Linear under autocast(bf16) for the transformer blocks: Y[M][N] = act(X[M][K] . W[N][K]^T + bias[N]).
 * Replaces the library GEMM behind F.linear at: timm's ViT blocks reached through models/vae.py:47-53 (attn.qkv / attn.proj / mlp.fc1 + nn.GELU /
 * mlp.fc2 and the patch embedding as a GEMM over patches), diffusion/lightningdit/lightningdit.py:34-93,173-252 (attn.qkv, attn.proj),
 * diffusion/lightningdit/swiglu_ffn.py:15-36 (w12, w3); with W := a transposed bf16 copy of the weight it is their input gradient dX = dY . W.
 * x [M][lda], w [N][ldw], y [M][ldy] row-major bf16 (y f32 when out_f32); leading dimensions in elements, multiples of 8; K % 32 == 0, N % 8 == 0;
 * every operand below 2 GiB.  w_layout = 1: w is the K-tile-major copy [K / 32][N][32] that dmvae_pack_conv_weight_v2 (ks = 1) writes as out_kmajor -- a
 * K tile of the weights is then one contiguous run of whole 128-B lines instead of N half lines a row apart, which is what the operand costs when it comes
 * from HBM rather than from the Infinity Cache (fc2 of ViT-L with cold operands: 84 -> 74 us; tools/bench_gemm.py --cold).  bias: f32 [N], or bf16 [N] when bias_bf16 (what autocast hands the library), or NULL.
 * act: 0 none, 1 SiLU, 5 exact (erf) GELU -- applied to the bf16-ROUNDED pre-activation, so the result is bit-identical to this call with act = 0
 * followed by dmvae_gelu_fwd / dmvae_silu_fwd.  act 6 = SwiGLU (swiglu_ffn.py:32-35: x1, x2 = w12(x).chunk(2, -1); silu(x1) * x2): N = 2 H weight rows [x1 | x2]
 * (N % 16 == 0, bf16 result only), y is [M][ldy >= H] and receives silu(x1) * x2 of the bf16-rounded halves -- bit-identical to act = 0 followed by
 * dmvae_swiglu_bf16, without x12 ever reaching HBM (the no-grad LightningDiT forward; the training route keeps x12 for its backward).  K >= 384 (the kernel streams a tile's last K steps together with the next tile's first ones, up to five
 * each); shorter reductions go to dmvae_gemm_nt_batched.
 * Tile shape per (M, N, K) from a fixed menu by rounds x tile cost (dmvae_linear_bf16_plan returns the menu index and the tile's columns / rows);
 * results do not depend on the tile (one f32 accumulation chain per output element: bias first, then K order).  csrc/gemm_pp.hip. */
int dmvae_linear_bf16(const void* x, const void* w, const void* bias, void* y, int M, int N, int K, int lda, int ldw, int ldy,
                      int act, int bias_bf16, int out_f32, int w_layout, dmvae_stream_t stream);
int dmvae_linear_bf16_plan(int M, int N, int K, int* tile_cols, int* tile_rows);
/* `batch` independent products y_b [M][ldy] = x_b [M][lda] w_b [N][ldw]^T on the same kernel (a batch index in its tile decode): the decoder attention's per-sample
 * GEMMs (flux_ae.py:37-49: q k^T, p v and their input gradients -- 1024 x 1024 x 512 per sample, 8-16 tiles of 256 x 256 each).  bf16 row-major operands, f32
 * accumulation in K order, bf16 or f32 result; sx / sw / sy: element strides between the products; no bias, no activation.  K % 32 == 0, K >= 384, N % 8 == 0,
 * M >= 64, every operand below 2 GiB over the whole batch. */
int dmvae_linear_bf16_batched_supported(int batch, int M, int N, int K);
int dmvae_linear_bf16_batched(const void* x, const void* w, void* y, int batch, int M, int N, int K, int lda, int ldw, int ldy, long long sx, long long sw,
                              long long sy, int out_f32, dmvae_stream_t stream);
/* Split-K form of dmvae_linear_bf16 for few-tile, deep-K problems (LightningDiT at batch 16: M = 4096 x N = 1152 is 18 tiles of 256 x 256; K = 3072 ... 6144):
 * the reduction cut into `splits` equal parts computed as independent work units into f32 slabs [splits][M][N] (no bias, no activation), then
 * dmvae_splitk_sum_bf16: y bf16 [M][N] = bf16(slab_0 + slab_1 + ... + bias) in that order (bias f32, or bf16 when bias_bf16; may be NULL).
 * K % (32 splits) == 0, K / splits >= 384, N % 8 == 0; w row-major [N][ldw] (w_layout 0) or K-tile-major (1).  Deterministic.  Reference: LightningDiTBlock's Linears, diffusion/lightningdit/lightningdit.py:173-252, swiglu_ffn.py:15-36. */
int dmvae_linear_bf16_splitk_supported(int M, int N, int K, int splits);
int dmvae_linear_bf16_splitk(const void* x, const void* w, void* slabs, int splits, int M, int N, int K, int lda, int ldw, int w_layout, dmvae_stream_t stream);
int dmvae_splitk_sum_bf16(const void* slabs, int splits, const void* bias, int bias_bf16, void* y, int M, int N, dmvae_stream_t stream);

/* SwiGLU FFN's first half in one launch: x12 [M][ldx12] = bf16(x w^T + bias) over the N = 2 H columns [x1 | x2] AND g [M][ldg] = silu(x1) * x2 (H columns): the
 * bits of dmvae_linear_bf16 (act 0) followed by dmvae_swiglu_bf16, without the second pass over the 2 H-wide tensor; x12 is what SwiGLU's backward reads.
 * N % 16 == 0.  Reference: swiglu_ffn.py:31-36 (w12, chunk, silu(x1) * x2). */
int dmvae_linear_bf16_swiglu_pre(const void* x, const void* w, const void* bias, void* g, void* x12, int M, int N, int K, int lda, int ldw, int ldg,
                                 int ldx12, int bias_bf16, int w_layout, dmvae_stream_t stream);

/* Stream-K / fused split-K Linear: y bf16 [M][ldy] = act(x [M][lda] w^T + bias), the reduction cut ACROSS workgroups and summed in K order by the last part of a
 * tile to arrive -- one launch, no slab pass (csrc/gemm_pp.hip, SK instantiation; 256 x 256 tiles).  For few-tile deep-K problems (LightningDiT-XL/1 at batch 16:
 * M 4096 x N 1152 is 80 tiles for 256 CUs; K = 3072 .. 6144) and problems of 1.x rounds of tiles.  splits = 0: stream-K -- 256 equal ranges of the flattened
 * (tile, K step) space, the cut depends on M; splits 2 .. 8: that many uniform parts per tile -- the cut depends on N and K only, so a row's bits do not depend
 * on the number of rows in the call (train_dmd.py:212-217 evaluated as one 2B call = two B calls).  Run-to-run identical either way (fixed summation order).
 * tile 0: 256 x 256 output tiles; tile 1: 256 columns x 128 rows (a third of the partial bytes per cut; what the few-tile shapes take).
 * act / bias / w_layout as dmvae_linear_bf16.  workspace >= dmvae_linear_bf16_sk_workspace(M, N, K, splits, tile) bytes whose FIRST dmvae_linear_bf16_sk_counter_bytes()
 * bytes are zero on entry (arrival counters; the kernel leaves them zero: zero the buffer once).  Reference: nn.Linear under autocast,
 * diffusion/lightningdit/lightningdit.py:66-75,236-250, swiglu_ffn.py:15-36, train_dmd.py:563-575 (their backward: dX = dY W). */
int dmvae_linear_bf16_sk_supported(int M, int N, int K, int splits, int tile);
size_t dmvae_linear_bf16_sk_counter_bytes(void);
size_t dmvae_linear_bf16_sk_workspace(int M, int N, int K, int splits, int tile);
int dmvae_linear_bf16_sk(const void* x, const void* w, const void* bias, void* y, void* workspace, size_t workspace_bytes, int splits, int tile,
                         int M, int N, int K, int lda, int ldw, int ldy, int act, int bias_bf16, int w_layout, dmvae_stream_t stream);
/* The input-gradient operand of dmvae_linear_bf16 from a Linear weight's bf16 copy: w bf16 [N][K] row-major (nn.Linear.weight: N = out_features, K = in_features)
 * -> out bf16 [N / 32][K][32], out[n >> 5][k][n & 31] = w[n][k], i.e. the K-tile-major layout (w_layout = 1) of W^T [K][N] with the reduction over n:
 * dmvae_linear_bf16(dy [M][N], out, NULL, dx, M, K, N, ..., w_layout = 1) is dX = dY . W.  N % 32 == 0, K % 8 == 0.  One tiled-transpose launch per weight
 * version on the routes where the weight trains (timm blocks via models/vae.py:47-53 with the encoder trainable, train_dmd.py:519; LightningDiT student,
 * train_dmd.py:565-575, train_diffusion.py:290-297). */
int dmvae_linear_weight_t_kmajor(const void* w, void* out, int N, int K, dmvae_stream_t stream);

/* P = softmax(scale*S) per row (S f32 [rows][cols] -> P bf16), and its backward
 * dS = scale * P .* (dP - rowsum(dP .* P)) (dP f32, dS bf16).  F.scaled_dot_product_attention's
 * softmax at flux_ae.py:47 (single head, scale = 1/sqrt(C)). */
int dmvae_softmax_rows_fwd(const void* s, void* p, int rows, int cols, float scale, dmvae_stream_t stream);
int dmvae_softmax_rows_bwd(const void* dp, const void* p, void* ds, int rows, int cols, float scale,
                           dmvae_stream_t stream);
/* dst[b][c][r] = src[b][r][c], bf16.  Reference: Operand layout for AttnBlock's products, models/flux_ae.py:37-49. */
int dmvae_transpose_bf16(const void* src, void* dst, int batch, int rows, int cols, dmvae_stream_t stream);

/* ---- layout / packing ------------------------------------------------------------------------- */

/* nn.Conv2d.weight f32 [cout][cin][ks][ks] -> bf16 kernel operand.
 * for_dgrad=0: out[rows_pad=cout..][ks*ks][cols_pad=cin..], out[co][t][ci]       = w[co][ci][t]
 * for_dgrad=1: out[rows_pad=cin..][ks*ks][cols_pad=cout..], out[ci][T-1-t][co]   = w[co][ci][t]
 * (tap-flipped transpose: conv2d_nhwc_fwd on dy with this operand is the input gradient).
 * Padding rows/cols are zero-filled.  Reference: The bf16 copy torch.autocast makes of the weight of every nn.Conv2d of models/flux_ae.py:21-107,239-269 and of the LPIPS trunk utils/lpips.py:116-153 on every forward. */
int dmvae_pack_conv_weight(const void* w, void* out, int cout, int cin, int ks, int rows_pad, int cols_pad,
                           int for_dgrad, dmvae_stream_t stream);
/* The same pack, and in the same launch the K-tile-major copy out_kmajor[cols_pad/32][ks*ks][rows_pad][32] (dmvae_conv_desc.w_layout = 1) when
 * out_kmajor is non-NULL (cols_pad % 32 == 0 then).  Reference: The bf16 copy torch.autocast makes of the weight of every nn.Conv2d of models/flux_ae.py:21-107,239-269 and of the LPIPS trunk utils/lpips.py:116-153 on every forward. */
int dmvae_pack_conv_weight_v2(const void* w, void* out, void* out_kmajor, int cout, int cin, int ks, int rows_pad, int cols_pad,
                              int for_dgrad, dmvae_stream_t stream);
/* Every stale weight operand of a model in ONE launch: after an optimiser step (train_tokenizer.py:416-417 `optimizer_vae.step()`) the bf16 operands of all
 * trainable conv weights have to be rewritten; one dmvae_pack_conv_weight_v2 launch per weight and direction is ~84 launches of 6-15 us per step for the
 * tokenizer.  `table` is a DEVICE array of n_entries dmvae_pack_entry (dmvae_pack_entry_bytes() each): entry e is one dmvae_pack_conv_weight_v2 call
 * (src, dst, dst2 = out_kmajor or NULL, cout / cin / T = ks*ks of the f32 tensor being packed, rows_pad, cols_pad, mode = for_dgrad) -- with subpixel = 1 the
 * tensor being packed is the sub-pixel weight WD [cin_w][cout_w][4][4] of dmvae_subpixel_weight, computed on the fly from src = W [cout_w][cin_w][3][3]
 * (cout = cin_w, cin = cout_w, T = 16) -- and owns blocks [start, start + count) of the launch, count = ceil(rows_pad / 32) * ceil(cols_pad / 32) (one block
 * packs a 32 x 32 tile of (row, column) pairs with all taps through LDS), starts ascending and contiguous from 0; total = their sum; max_taps = the largest number of taps of the memory tensors (T, 9 for a sub-pixel entry) <= 16.  Results are
 * bit-identical to the per-weight calls. */
typedef struct {
  const float* src; void* dst; void* dst2;
  int32_t cout, cin, T, rows_pad, cols_pad, mode, subpixel, reserved;
  uint64_t start, count;
} dmvae_pack_entry;
size_t dmvae_pack_entry_bytes(void);
int dmvae_pack_weights_batched(const void* table, int n_entries, unsigned long long total, int max_taps, dmvae_stream_t stream);
/* Sub-pixel form of Upsample's conv (models/flux_ae.py:103-107: conv3x3(F.interpolate(x, 2, 'nearest'))): output pixel (2y+py, 2x+px) only sees the 2x2
 * source pixels around (y, x), so taps that land on one source pixel are added up front --
 *   conv2d(interpolate(x,2), W, padding=1) == conv_transpose2d(x, WD, stride=2, padding=1),
 *   WD[ci][co][r][s] = sum_{ky in K(r), kx in K(s)} W[co][ci][ky][kx],  K(r) = {k : 2 <= r+k <= 3}
 * i.e. 16 taps per source pixel instead of 9 per output pixel (4/9 of the multiply-adds).  With D = the 4x4 stride-2 padding-1 conv whose weight is WD
 * (Cout -> Cin):  forward  = conv2d_nhwc_fwd(x, pack(WD, for_dgrad=1), desc{ks 4, stride 2, transposed 1});
 *                 dL/dx    = conv2d_nhwc_fwd(dy, pack(WD), desc{ks 4, stride 2});
 *                 dL/dWD   = conv2d_nhwc_wgrad(dy := x, a := dy, desc{ks 4, stride 2}), dL/dW = subpixel_weight_fold(dL/dWD);
 *                 dL/dbias = colsum_bf16(dy).
 * w: f32 [cout][cin][3][3]; wd: f32 [cin][cout][4][4]. */
int dmvae_subpixel_weight(const void* w, void* wd, int cout, int cin, dmvae_stream_t stream);
/* dw[co][ci][ky][kx] (+)= sum_{r in {2-ky,3-ky}, s in {2-kx,3-kx}} dwd[ci][co][r][s] -- the transpose of the map above (f32, fixed order).  Reference: Upsample = nearest x2 + conv3x3, models/flux_ae.py:98-107. */
int dmvae_subpixel_weight_fold(const void* dwd, void* dw, int cout, int cin, int accumulate, dmvae_stream_t stream);
/* out[c] (+)= sum_r x[r][c], x row-major bf16 [rows][cols], out f32 [cols]; two-stage, fixed order.  workspace >= 512*cols*4 bytes.
 * The bias gradient of nn.Conv2d on its own (autograd's sum over N,H,W of the output gradient).  Reference: Bias gradient of every nn.Conv2d of models/flux_ae.py:21-107,239-269 and of the LPIPS trunk utils/lpips.py:116-153. */
int dmvae_colsum_bf16(const void* x, void* out, void* workspace, size_t workspace_bytes, size_t rows, int cols, int accumulate,
                      dmvae_stream_t stream);
/* dx[n,h,w,c] = sum of the 2x2 block of dy[n,2h,2w,c]: backward of F.interpolate(scale=2,'nearest')
 * (flux_ae.py:104).  bf16, c%8==0. */
int dmvae_sumpool2x2_nhwc(const void* dy, void* dx, int n, int h, int w, int c, dmvae_stream_t stream);
/* PatchGAN convs (models/patchgan.py:125-147: nn.Conv2d(k=4, stride 2 or 1, padding 1)) as im2col + the 1x1 GEMM path:
 * col[n,oy,ox,(ky*ks+kx)*c + ci] = x[n, oy*stride-pad+ky, ox*stride-pad+kx, ci] (zero outside), ho = (h+2*pad-ks)/stride+1;
 * col2im is its adjoint (f32 accumulation over the overlapping taps, gather form: deterministic; dcol bf16, or f32 when
 * in_f32 -- the GEMM's f32 result summed without an intermediate rounding, as a direct dgrad conv would).  bf16 x/col/dx, c%8==0.
 * The weight operand [cout][ks*ks][c] comes from dmvae_pack_conv_weight (ks up to 7). */
int dmvae_im2col_nhwc(const void* x, void* col, int n, int h, int w, int c, int ks, int stride, int pad, dmvae_stream_t stream);
/* The same with the tap count padded to taps_pad >= ks * ks: col is [n, ho, wo, taps_pad * c], taps past the last one are columns of zeros -- a 3x3 conv over 32
 * channels becomes a 384-column operand (12 x 32), which the 1x1 weight-gradient kernel's 128-column tiles take (the decoder's conv_in, flux_ae.py:196). */
int dmvae_im2col_nhwc_taps(const void* x, void* col, int n, int h, int w, int c, int ks, int stride, int pad, int taps_pad, dmvae_stream_t stream);
/* The same from the first c of c_src channels of every source pixel (c_src % 8 == 0, c_src >= c): the 8-channel im2col of the first PatchGAN layer's 32-channel
 * padded input (models/patchgan.py:125) without a sliced copy in between. */
int dmvae_im2col_nhwc_sub(const void* x, void* col, int n, int h, int w, int c_src, int c, int ks, int stride, int pad, int taps_pad, dmvae_stream_t stream);
int dmvae_col2im_nhwc(const void* dcol, void* dx, int n, int h, int w, int c, int ks, int stride, int pad, int in_f32,
                      dmvae_stream_t stream);
/* dx = y > 0 ? dy : slope*dy  (nn.LeakyReLU(0.2) backward from the saved OUTPUT, patchgan.py:125,136,144). bf16, n%8==0. */
int dmvae_leaky_relu_bwd(const void* dy, const void* y, void* dx, size_t n, float slope, dmvae_stream_t stream);
/* nn.BatchNorm2d's running-estimate update (models/patchgan.py:125-147's norm layers in training mode) from the batch statistics this build's kernels keep as
 * (mean, rstd) pairs, stats f32 [c][2]: var = max(1 / rstd^2 - eps, 0); running_mean = (1 - momentum) running_mean + momentum mean; running_var = (1 - momentum)
 * running_var + momentum * unbias * var (unbias = n / (n - 1)).  f32 running buffers, in place. */
int dmvae_batchnorm_running_update(const void* stats, void* running_mean, void* running_var, int c, float eps, float momentum, float unbias, dmvae_stream_t stream);
/* DiffAug (utils/diffaug.py:43-114) on NCHW f32 images [b,c<=8,h,w], blur warm-up off (schedule 0 at every reference call site):
 * translate by th,tw = floor(r0|r1*(2*delta+1)) - delta with zero fill; brightness r2-0.5, per-pixel saturation 2*r3, per-image
 * contrast r4+0.5; zero the cut_h x cut_w rectangle centred at floor(r5|r6*(size+1-cut%2)), clamped at the border.
 * rand01: device f32 [7][b] = torch.rand(7,b,1,1) (:69); flags: bit 0 translate, bit 1 colour, bit 2 cut-out (torch.rand(3) <= prob,
 * :66); delta_* = round(size*0.125), cut_* = round(size*cutout) (:73-75,:92-94).  workspace: 36 * b floats, 8-byte aligned (the b per-image means + the
 * two-stage reduction's f64 partials).  bwd is the exact adjoint. */
int dmvae_diffaug_fwd(const void* x, const void* rand01, void* y, void* workspace, int b, int c, int h, int w, int flags,
                      int delta_h, int delta_w, int cut_h, int cut_w, dmvae_stream_t stream);
int dmvae_diffaug_bwd(const void* dy, const void* rand01, void* dx, void* workspace, int b, int c, int h, int w, int flags,
                      int delta_h, int delta_w, int cut_h, int cut_w, dmvae_stream_t stream);
/* VGG16 trunk of LPIPS (utils/lpips.py:116-153; nn.MaxPool2d(2,2) and the ReLU backward), NHWC bf16, c%8==0:
 * y[n,h,w,c] = max of the 2x2 window of x[n,2h,2w,c];
 * dx = x > 0 ? route(dpool -> first maximum of its window) + extra : 0   (dpool [n,h,w,c] and extra [n,2h,2w,c] may be NULL);
 * relu_bwd: dx = y > 0 ? dy : 0. */
int dmvae_maxpool2x2_nhwc(const void* x, void* y, int n, int h, int w, int c, dmvae_stream_t stream);
int dmvae_maxpool2x2_relu_bwd_nhwc(const void* dpool, const void* x, const void* extra, void* dx, int n, int h, int w, int c,
                                   dmvae_stream_t stream);
int dmvae_relu_bwd(const void* dy, const void* y, void* dx, size_t n, dmvae_stream_t stream);
/* image layout conversion at the Decoder boundary (reference tensors are NCHW f32).  Reference: models/vae.py:56-65 (decode / forward hand NCHW f32 images to and from the Decoder). */
int dmvae_nchw_f32_to_nhwc_bf16(const void* src, void* dst, int n, int c, int hw, int c_pad, dmvae_stream_t stream);
int dmvae_nhwc_to_nchw_f32(const void* src, void* dst, int n, int c, int hw, int c_pad, int src_f32,
                           dmvae_stream_t stream);
/* SiLU on bf16 (nn.SiLU in the bottleneck MLP, vae.py:60) and its backward. n%8==0. */
int dmvae_silu_fwd(const void* x, void* y, size_t n, dmvae_stream_t stream);
int dmvae_silu_bwd(const void* x, const void* dy, void* dx, size_t n, dmvae_stream_t stream);

/* ---- frozen ViT encoder forward, elementwise part (models/vae.py:52-53; timm / dino_layers block algebra) ------------------- */

/* y[rows][c] (bf16) = LayerNorm(x[rows][c] f32; gamma, beta f32, eps): nn.LayerNorm under autocast (f32) + the bf16 cast in front
 * of the following Linear.  c in {256, 512, ..., 1536}.  Reference: timm VisionTransformer blocks' norm1 / norm2 / norm through models/vae.py:47-53. */
int dmvae_layernorm_f32_bf16(const void* x, const void* gamma, const void* beta, void* y, int rows, int c, float eps,
                             dmvae_stream_t stream);
/* The two calls below fused, for every LayerNorm that follows a LayerScale + residual add (all but a block's first): x += ls_gamma * r (in place, f32), then
 * y = bf16(LayerNorm(x)); one pass over the residual stream, bit-identical to scale_residual_f32 followed by layernorm_f32_bf16.  Reference: timm Block: x + ls(attn(norm1(x))), x + ls(mlp(norm2(x))) through models/vae.py:47-53; LayerScale models/dino_layers/layer_scale.py:18-27. */
int dmvae_scale_residual_layernorm(void* x, const void* r, const void* ls_gamma, const void* gamma, const void* beta, void* y, int rows, int c, float eps,
                                   dmvae_stream_t stream);
/* x[rows][c] (f32, in place) += gamma[c] * y[rows][c] (bf16): LayerScale (dino_layers/layer_scale.py:15-26) + residual add. c%8==0. */
int dmvae_scale_residual_f32(void* x, const void* y, const void* gamma, size_t rows, int c, dmvae_stream_t stream);
/* p[rows][cols] (bf16) = softmax(scale * s[rows][cols]) with bf16 scores, f32 inside; cols <= 512 (encoder attention, S = 257).  Reference: models/dino_layers/attention.py:56-69 (unfused path). */
int dmvae_softmax_rows_bf16(const void* s, void* p, size_t rows, int cols, float scale, dmvae_stream_t stream);
/* out[b][s][h*64+d] = softmax_k(scale * q.k) v for qkv [b][s][3][h][64] bf16 (the qkv Linear's output layout): the encoder's
 * multi-head self-attention (timm Attention; dino_layers/attention.py:56-69) fused in one kernel.  head_dim 64, seq <= 288. */
int dmvae_attention_qkv_bf16(const void* qkv, void* out, int batch, int seq, int heads, int head_dim, float scale,
                             dmvae_stream_t stream);
/* The same fused kernel on head-major operands: q, k [batch*heads][seq][head_dim_padded] (channels >= head_dim zero), v
 * [batch*heads][seq][head_dim] bf16 -- what dmvae_qknorm_rope_bf16 produces -> out [batch][seq][heads*head_dim].  LightningDiT's
 * attention after QK-norm + RoPE (models/lightningdit.py:64-98, F.scaled_dot_product_attention); head_dim % 8 == 0, seq <= 288; head_dim_padded = the channels
 * a q / k row holds: 64 or 96, or head_dim itself (<= 96; rows without padding: the kernel does not touch the chunks past head_dim and computes the same bits). */
int dmvae_attention_heads_bf16(const void* q, const void* k, const void* v, void* out, int batch, int seq, int heads, int head_dim,
                               int head_dim_padded, float scale, dmvae_stream_t stream);
/* The two kernels above with the row statistics written out: lse f32 [batch * heads][seq] = scale * max_k(q.k) + log(sum_k exp(scale (q.k - max))) per query -- what
 * dmvae_attention_bwd_*_lse_bf16 rebuild the probabilities from (lse may be NULL: the plain calls).  Reference: models/dino_layers/attention.py:56-69; diffusion/lightningdit/lightningdit.py:76-88 (F.scaled_dot_product_attention). */
int dmvae_attention_qkv_lse_bf16(const void* qkv, void* out, void* lse, int batch, int seq, int heads, int head_dim, float scale, dmvae_stream_t stream);
int dmvae_attention_heads_lse_bf16(const void* q, const void* k, const void* v, void* out, void* lse, int batch, int seq, int heads, int head_dim,
                                   int head_dim_padded, float scale, dmvae_stream_t stream);
/* The whole attention of a LightningDiT block from the qkv Linear's output [batch][seq][3][heads][head_dim] bf16: per-head RMSNorm (bf16 result) * weight
 * and the 2-D rotary embedding (the arithmetic of dmvae_qknorm_rope_bf16; cos / sin tables [seq][head_dim] f32) are applied to q and k as they enter the
 * fused kernel -> out [batch][seq][heads*head_dim].  lightningdit.py:66-88 in one launch, no head-major q / k / v in HBM.  head_dim % 8 == 0, <= 96; seq <= 288. */
int dmvae_attention_qknorm_rope_bf16(const void* qkv, const void* q_weight, const void* k_weight, const void* cos_table, const void* sin_table,
                                     void* out, int batch, int seq, int heads, int head_dim, float eps, float scale, dmvae_stream_t stream);

/* Backward of the fused attention above (autograd's SDPA backward for timm Attention, dino_layers/attention.py:56-69, in the stages where the encoder
 * trains -- train_dmd.py:349,519 -- and for LightningDiT's Attention, lightningdit.py:76-88, in the student's training turn): one kernel per call, the
 * S x S probabilities are recomputed in registers and never reach HBM.  out = the forward result [batch][seq][heads*head_dim], dout = its gradient (bf16).
 * _qkv: qkv [batch][seq][3][heads][64] -> dqkv in the same layout (every element of rows < seq written).  head_dim 64, seq <= 288.
 * _heads: q, k [batch*heads][seq][head_dim_padded], v [batch*heads][seq][head_dim] -> dq, dk, dv in the same layouts (padded channels of dq / dk
 *         come out as the zeros the padded operands imply).  head_dim % 8 == 0, head_dim_padded 64, 96 or head_dim (rows without padding), seq <= 288. */
int dmvae_attention_bwd_qkv_bf16(const void* qkv, const void* out, const void* dout, void* dqkv, int batch, int seq, int heads, int head_dim,
                                 float scale, dmvae_stream_t stream);
int dmvae_attention_bwd_heads_bf16(const void* q, const void* k, const void* v, const void* out, const void* dout, void* dq, void* dk, void* dv,
                                   int batch, int seq, int heads, int head_dim, int head_dim_padded, float scale, dmvae_stream_t stream);
/* The same with the forward's row statistics handed in (lse from dmvae_attention_*_lse_bf16; NULL = the calls above): the probabilities are rebuilt as
 * exp(scale q.k - lse) without a max / sum pass, which frees the registers for two waves per SIMD (eight-wave workgroups).  Reference: Autograd of models/dino_layers/attention.py:56-69 and diffusion/lightningdit/lightningdit.py:76-88 (train_dmd.py:565-575). */
int dmvae_attention_bwd_qkv_lse_bf16(const void* qkv, const void* out, const void* dout, const void* lse, void* dqkv, int batch, int seq, int heads, int head_dim,
                                     float scale, dmvae_stream_t stream);
int dmvae_attention_bwd_heads_lse_bf16(const void* q, const void* k, const void* v, const void* out, const void* dout, const void* lse, void* dq, void* dk,
                                       void* dv, int batch, int seq, int heads, int head_dim, int head_dim_padded, float scale, dmvae_stream_t stream);

/* Backward side of the same encoder block, for the stages where the encoder trains (train_dmd.py:349,519).  Residual stream f32,
 * Linear operands / results bf16 (autocast).  workspace: dmvae_vit_bwd_workspace(c) bytes.
 * layernorm_bwd: dx_io[rows][c] (f32, the residual-stream gradient) += LayerNorm backward of dy (bf16) at input x (f32);
 *   dgamma / dbeta ([c] f32, both or neither; accumulate != 0 adds).  c as for layernorm_f32_bf16.
 * layerscale_bwd: for r = x + gamma * y: dy = gamma * dt (bf16), dgamma (+)= sum_rows dt * y.  c = 8 * a divisor of 256.
 * gelu: exact erf form (nn.GELU() default), bf16 in / out, f32 inside; bwd takes the pre-activation x.  n % 8 == 0. */
size_t dmvae_vit_bwd_workspace(int c);
int dmvae_layernorm_bwd_f32(const void* dy, const void* x, const void* gamma, void* dx_io, void* dgamma, void* dbeta, void* workspace,
                            size_t workspace_bytes, int rows, int c, float eps, int accumulate, dmvae_stream_t stream);
int dmvae_layerscale_bwd(const void* dt, const void* y, const void* gamma, void* dy, void* dgamma, void* workspace,
                         size_t workspace_bytes, int rows, int c, int accumulate, dmvae_stream_t stream);
int dmvae_gelu_fwd(const void* x, void* y, size_t n, dmvae_stream_t stream);
int dmvae_gelu_bwd(const void* dy, const void* x, void* dx, size_t n, dmvae_stream_t stream);

/* ---- LightningDiT inference path (diffusion/lightningdit/lightningdit.py:175-273; teacher / student evaluations of the DMD loss,
 * train_dmd.py:211-217).  Residual stream f32, Linear operands / results bf16; rounding sites follow the reference's autocast graph.
 * mod: the adaLN Linear's output [B][mod_stride] bf16; *_off: element offsets of the shift / scale / gate chunks inside a row.
 * rmsnorm_modulate: y = bf16( RMSNorm(x; w, eps) * bf16(1 + scale[b]) + shift[b] ), shift_off < 0: no shift (`wo_shift`).  c % 4 == 0, c <= 2048.
 * qknorm_rope: qkv [B][N][3][H][D] bf16 -> q, k: per-head RMSNorm (bf16 result) * weight, 2-D rotary embedding with the [N][D] cos / sin
 *   tables (pos_embed.py:96-135), bf16, head-major [B*H][N][Dp] zero-padded to Dp >= D; v copied head-major [B*H][N][D].  D even, Dp <= 128.
 * swiglu: out[rows][hidden] = bf16( bf16(silu(x1)) * x2 ), x12 = [x1 | x2] (swiglu_ffn.py:31-36).  hidden % 8 == 0.
 * gated_residual: x[rows][c] (f32) += bf16( gate[b] * y ).  c % 8 == 0.
 * gated_residual_rmsnorm_modulate: the two back to back in one pass over the residual stream -- x += bf16(gate[b] * r) (gate from gate_mod, which may
 *   be another adaLN output than `mod`: the next block's), x written back, then y = rmsnorm_modulate(x). */
int dmvae_rmsnorm_modulate_bf16(const void* x, const void* w, const void* mod, void* y, int rows, int rows_per_sample, int c, int mod_stride,
                                int shift_off, int scale_off, float eps, dmvae_stream_t stream);
int dmvae_gated_residual_rmsnorm_modulate(void* x, const void* r, const void* gate_mod, int gate_stride, int gate_off, const void* w,
                                          const void* mod, void* y, int rows, int rows_per_sample, int c, int mod_stride, int shift_off,
                                          int scale_off, float eps, dmvae_stream_t stream);
/* Out-of-place form (the training route keeps the residual stream before the update for its backward): x_out = x_in + bf16(gate[b] * r); y != NULL: also
 * y = rmsnorm_modulate(x_out) in the same pass (w, mod and the offsets are then required), y == NULL: the residual update alone.  Reference: diffusion/lightningdit/lightningdit.py:27-31,66-75,236-250; swiglu_ffn.py:32-35; rms_norm.py:52-76. */
int dmvae_gated_residual_out(const void* x_in, void* x_out, const void* r, const void* gate_mod, int gate_stride, int gate_off, const void* w,
                             const void* mod, void* y, int rows, int rows_per_sample, int c, int mod_stride, int shift_off, int scale_off,
                             float eps, dmvae_stream_t stream);
int dmvae_qknorm_rope_bf16(const void* qkv, const void* q_weight, const void* k_weight, const void* cos_table, const void* sin_table,
                           void* q_out, void* k_out, void* v_out, int batch, int seq, int heads, int head_dim, int head_dim_padded, float eps,
                           dmvae_stream_t stream);
int dmvae_swiglu_bf16(const void* x12, void* out, size_t rows, int hidden, dmvae_stream_t stream);
int dmvae_gated_residual_f32(void* x, const void* y, const void* mod, size_t rows, int rows_per_sample, int c, int mod_stride, int gate_off,
                             dmvae_stream_t stream);
/* Backward side of the four kernels above (the student's training turn, train_dmd.py:565-575).  dmod: f32 [B][mod_stride], each call fills the
 * chunk(s) it owns (d shift / d scale / d gate = sums over the sample's tokens).  workspace: dmvae_dit_bwd_workspace(batch, c) bytes.
 * gated_residual_bwd: dy = bf16(gate[b] * dx), dmod[gate] = sum_n dx * y.
 * swiglu_bwd: dx12 = [dh * x2 * silu'(x1) | dh * bf16(silu(x1))].
 * rmsnorm_modulate_bwd: dx_io (f32) += RMSNorm backward of da * w * bf16(1 + scale); dmod[shift] = sum_n da (skipped when shift_off < 0),
 *   dmod[scale] = sum_n da * n * w, dw (+)= sum_rows da * bf16(1 + scale) * n   (dw may be NULL).
 * qknorm_rope_bwd: (dq, dk [B*H][N][Dp], dv [B*H][N][D]) -> dqkv [B][N][3][H][D] bf16 through the transposed rotation and the per-head RMSNorm;
 *   dq_weight / dk_weight [D] f32 (accumulate != 0 adds). */
size_t dmvae_dit_bwd_workspace(int batch, int c);
int dmvae_gated_residual_bwd(const void* dx, const void* y, const void* mod, void* dy, void* dmod, int batch, int seq, int c, int mod_stride,
                             int gate_off, dmvae_stream_t stream);
int dmvae_swiglu_bwd(const void* dh, const void* x12, void* dx12, size_t rows, int hidden, dmvae_stream_t stream);
int dmvae_rmsnorm_modulate_bwd(const void* da, const void* x, const void* w, const void* mod, void* dx_io, void* dmod, void* dw, void* workspace,
                               size_t workspace_bytes, int batch, int seq, int c, int mod_stride, int shift_off, int scale_off, float eps,
                               int accumulate, dmvae_stream_t stream);
int dmvae_qknorm_rope_bwd(const void* dq, const void* dk, const void* dv, const void* qkv, const void* q_weight, const void* k_weight,
                          const void* cos_table, const void* sin_table, void* dqkv, void* dq_weight, void* dk_weight, void* workspace,
                          size_t workspace_bytes, int batch, int seq, int heads, int head_dim, int head_dim_padded, float eps, int accumulate,
                          dmvae_stream_t stream);

/* First stage of dmvae_qknorm_rope_bwd only: dqkv is written, the per-block partial sums of the two norm-weight gradients stay in `part`
 * ([dmvae_qknorm_rope_bwd_nblk(...)][2][head_dim] f32) for a reduction batched over layers (dmvae_colsum2_batched).  Reference: Autograd of q_norm / k_norm + rotary embedding, diffusion/lightningdit/lightningdit.py:66-75 (train_dmd.py:565-575). */
int dmvae_qknorm_rope_bwd_nblk(int batch, int seq, int heads, int head_dim, int head_dim_padded);
int dmvae_qknorm_rope_bwd_partial(const void* dq, const void* dk, const void* dv, const void* qkv, const void* q_weight, const void* k_weight,
                                  const void* cos_table, const void* sin_table, void* dqkv, void* part, size_t part_bytes, int batch, int seq, int heads,
                                  int head_dim, int head_dim_padded, float eps, dmvae_stream_t stream);

/* Whole-stack backward of LightningDiT's blocks (diffusion/lightningdit/lightningdit.py:236-250 x depth; the student's flow-matching step, train_dmd.py:565-575,
 * train_diffusion.py:290-297) -- csrc/dit_stack.hip, driven by dmvae_amd/functional.py::DitStackFn.
 * dmvae_dit_boundary_bwd: one pass over the f32 residual-stream gradient dx_io [B][seq][c] at a sub-layer boundary.  Norm half (da != NULL): dx_io += backward of
 *   a = bf16(RMSNorm(x) * w * bf16(1 + scale[b]) + shift[b]) for the incoming da (bf16), x the stream the norm read (f32), mod bf16 [B][mod_stride] with the scale
 *   chunk at scale_off; per-sample partial sums of d shift, d scale, d w.  Gate half (y != NULL): dy = bf16(gate[b] * dx_io) and the partial sums of
 *   d gate[b] = sum_n dx_io * y for the gated residual x_out = x + bf16(gate * y) met NEXT on the way back (y its branch output, bf16; gate_mod bf16 [B][gate_stride]
 *   with the gate chunk at gate_off), computed from the UPDATED dx_io.  part_slot: this boundary's [B][dmvae_dit_stack_bps(B)][4][c] f32 slice of the stack's
 *   partial-sum array (slot order below); rowstat: B * seq float2 scratch (the tail of dmvae_dit_stack_workspace).
 * dmvae_dit_stack_finalize: after the last boundary, ONE reduction of all 2 L + 1 slots -- slot 2 l: norm1 of block l (+ the MLP gate of block l - 1 for l > 0),
 *   slot 2 l + 1: norm2 of block l + the attention gate of block l, slot 2 L: the MLP gate of block L - 1 alone -- into dmod bf16 [L][B][6 c] (adaLN chunk order:
 *   shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp) and the norm-weight gradients dw_table[2 l] (norm1 of block l) / [2 l + 1] (norm2): device array
 *   of 2 L pointers to f32 [c] (accumulate != 0 adds).  Two-stage, fixed order: deterministic. */
int dmvae_dit_stack_bps(int batch);
size_t dmvae_dit_stack_part_bytes(int layers, int batch, int c);
size_t dmvae_dit_stack_workspace(int layers, int batch, int seq, int c);
int dmvae_dit_boundary_bwd(const void* da, const void* x, const void* w, const void* mod, int mod_stride, int scale_off, float eps, void* dx_io, const void* y,
                           const void* gate_mod, int gate_stride, int gate_off, void* dy, void* part_slot, void* rowstat, int batch, int seq, int c,
                           dmvae_stream_t stream);
int dmvae_dit_stack_finalize(const void* part, void* dmod, void* workspace, size_t workspace_bytes, const void* dw_table, int layers, int batch, int seq, int c,
                             int accumulate, dmvae_stream_t stream);
/* part [layers][nblk][2][d] f32 -> o0_table[l][d], o1_table[l][d] (device arrays of `layers` pointers to f32 [d]): the second stage of
 * dmvae_qknorm_rope_bwd_partial for every layer in one launch.  Reference: Autograd of q_norm / k_norm weights, diffusion/lightningdit/lightningdit.py:66-75. */
int dmvae_colsum2_batched(const void* part, const void* o0_table, const void* o1_table, int layers, int nblk, int d, int accumulate, dmvae_stream_t stream);
/* dmvae_linear_rows_bf16 for `layers` Linears of one shape in one launch (adaLN_modulation[1] of every block: lightningdit.py:236-240): w_table / bias_table device
 * arrays of `layers` pointers (bias_table may be NULL), x / y advanced by x_layer_stride / y_layer_stride ELEMENTS per layer (x_layer_stride 0: one x for all). */
int dmvae_linear_rows_batched_bf16(const void* x, long long x_layer_stride, const void* w_table, const void* bias_table, void* y, long long y_layer_stride, int layers,
                                   int M, int N, int K, int ldx, int ldw, int ldy, int act, int bias_bf16, int out_f32, int w_layout, dmvae_stream_t stream);
/* Weight + bias gradients of `layers` per-sample Linears sharing their input, on the matrix cores: dW_l [N][K] f32 (+)= dY_l [M][N]^T . X [M][K], db_l [N] f32 (+)=
 * column sums of dY_l; dy bf16 [layers][M][lddy] (dy_layer_stride elements apart), xT = X TRANSPOSED, bf16 [K][mp] with mp = 32 or 64 >= M and zeros beyond M;
 * dw_table / db_table: device arrays of `layers` pointers (db_table may be NULL), or NULL tables with layers = 1 and dw / db given directly.  1 <= M <= 64, K % 8 == 0.
 * Bound by writing the f32 gradients; deterministic.  Reference: Autograd of every block's adaLN_modulation[1], diffusion/lightningdit/lightningdit.py:236-250. */
int dmvae_linear_rows_wgrad_batched(const void* dy, long long dy_layer_stride, const void* xT, int mp, const void* dw_table, const void* db_table, void* dw, void* db,
                                    int layers, int M, int N, int K, int lddy, int accumulate, dmvae_stream_t stream);
/* dmvae_linear_weight_t_kmajor for a table of weights in one launch (every Linear weight of a trainable transformer after its optimiser step).  table: device array of
 * n_entries records {const void* src; void* dst; int32 N, K; uint32 start, tiles_x} (dmvae_wt_entry_bytes() bytes each), entry e owning the flat tiles
 * [start_e, start_e + tiles_x * N / 32), tiles_x = ceil(K / 64); total_tiles = their sum.  Reference: The bf16 copy torch.autocast makes of every nn.Linear weight (diffusion/lightningdit/lightningdit.py:173-252; timm blocks via models/vae.py:47-53), as the input gradient's operand. */
size_t dmvae_wt_entry_bytes(void);
int dmvae_linear_weight_t_kmajor_batched(const void* table, int n_entries, unsigned total_tiles, dmvae_stream_t stream);

/* Grouped Linear weight gradients: ONE launch for a table of independent problems dW_p [cout_p][cin_p] f32 = dY_p [M_p][cout_p]^T . X_p [M_p][cin_p] (bf16 row-major
 * operands; nn.Linear's weight gradient under autocast), each problem unsplit -- its whole reduction in one workgroup per 256 x 256 output tile, written straight to dw
 * (no slabs, no reduce launch) -- plus their bias gradients db_p [cout_p] = column sums of dY_p.  For call sites that hold many at once: the 4 x 28 Linears of
 * LightningDiT's backward pass (diffusion/lightningdit/lightningdit.py:173-252, swiglu_ffn.py:15-36), the four of a ViT block (timm blocks via models/vae.py:47-53).
 * The table is built on the host, record by record, with dmvae_linear_wgrad_grouped_fill (entry: dmvae_linear_wgrad_grouped_entry_bytes() bytes; bias_entry:
 * ..._bias_entry_bytes() bytes, only for problems with db != NULL, which also need bias_part: f32 scratch of dmvae_linear_wgrad_grouped_bias_parts(cin) * cout
 * floats); *start / *bias_start accumulate the launch's block counts.  Then copy both tables to the device and call dmvae_linear_wgrad_grouped(table, n, total_blocks,
 * ragged, bias_table, n_bias, bias_blocks): ragged != 0 when some M_p is not a multiple of 32 (the last K tile's missing rows are masked per lane).
 * M >= 32, cout and cin multiples of 128, every operand below 2 GiB.  Deterministic (fixed accumulation order per tile). */
size_t dmvae_linear_wgrad_grouped_entry_bytes(void);
size_t dmvae_linear_wgrad_grouped_bias_entry_bytes(void);
int dmvae_linear_wgrad_grouped_supported(int M, int cout, int cin);
int dmvae_linear_wgrad_grouped_bias_parts(int cin);
int dmvae_linear_wgrad_grouped_fill(void* entry, void* bias_entry, const void* dy, const void* x, void* dw, void* bias_part, void* db, int M, int cout, int cin,
                                    unsigned* start, unsigned* bias_start);
int dmvae_linear_wgrad_grouped(const void* table, int n, unsigned total_blocks, int ragged, const void* bias_table, int n_bias, unsigned bias_blocks,
                               dmvae_stream_t stream);
/* The same launch with every tile placed on an XCD: dmvae_linear_wgrad_grouped_plan (host only) cuts the n problems of a filled table into chunks of whole cout-tile
 * rows of about 32 tiles -- one round of an XCD's 32 CUs -- and deals them to the eight XCDs, longest first to the least loaded; chunks_out receives *n_chunks
 * records of dmvae_linear_wgrad_grouped_chunk_bytes() (room for max_chunks; n * 64 is always enough), xoff[0..8] the XCDs' ranges in it, *grid the block count.
 * Copy the chunk records to the device too and launch with dmvae_linear_wgrad_grouped_xcd(table, chunks, xoff (host), grid, ragged, bias_table, n_bias, bias_blocks).
 * The tiles an XCD works on at a time then share their operand panels in that XCD's L2 (the first form spread each problem over all XCDs: 16 GB read where
 * the operands are 4.2 GB at LightningDiT-XL/1, batch 16).  Same tiles, same accumulation order per tile: same bits as dmvae_linear_wgrad_grouped.  Reference: diffusion/lightningdit/lightningdit.py:173-252, swiglu_ffn.py:15-36. */
size_t dmvae_linear_wgrad_grouped_chunk_bytes(void);
int dmvae_linear_wgrad_grouped_plan(const void* table, int n, void* chunks_out, int max_chunks, int* n_chunks, unsigned* xoff, unsigned* grid);
int dmvae_linear_wgrad_grouped_xcd(const void* table, const void* chunks, const unsigned* xoff, unsigned grid, int ragged, const void* bias_table, int n_bias,
                                   unsigned bias_blocks, dmvae_stream_t stream);

/* 4x4 stride-1 padding-1 convolution to ONE output channel: the PatchGAN's logits layer (models/patchgan.py:146, nn.Conv2d(8 ndf = 512, 1, 4, 1, 1)), forward and both
 * gradients on the vector units -- by its bytes the layer is one pass over the activation map (csrc/conv_c1.hip).  x, dx: NHWC bf16 [n][h][31][c]; w: the f32
 * parameter [1][c][4][4] (rounded to bf16 in registers: the autocast conv's operand), bias f32 [1] or NULL; out, dy: f32 [n][h - 1][30]; dw f32 [1][c][4][4] and
 * db f32 [1] (or NULL) are WRITTEN.  workspace: dmvae_conv_k4c1_wgrad_workspace(c) bytes.  Shapes: w == 31, c a multiple of 512 (dmvae_conv_k4c1_supported);
 * others run on dmvae_conv2d_nhwc_*.  f32 products and sums in a fixed order: deterministic. */
int dmvae_conv_k4c1_supported(int n, int h, int w, int c);
size_t dmvae_conv_k4c1_wgrad_workspace(int c);
int dmvae_conv_k4c1_fwd(const void* x, const void* w, const void* bias, void* out, int n, int h, int wdt, int c, dmvae_stream_t stream);
int dmvae_conv_k4c1_dgrad(const void* dy, const void* w, void* dx, int n, int h, int wdt, int c, dmvae_stream_t stream);
int dmvae_conv_k4c1_wgrad(const void* x, const void* dy, void* dw, void* db, void* workspace, size_t workspace_bytes, int n, int h, int wdt, int c,
                          dmvae_stream_t stream);

/* ---- losses (HBM-bound reductions) -------------------------------------------------------------  Reference: F.l1_loss / F.mse_loss / LPIPS of train_tokenizer.py:180-186. */

size_t dmvae_loss_workspace(void);

/* out2 = { mean|recon-images|, mean (recon-images)^2 } over n f32 elements (F.l1_loss, F.mse_loss,
 * train_tokenizer.py:180-181).  grad (optional, f32 [n]) = w1*sign(d)/n + w2*2d/n. */
int dmvae_l1_mse(const void* recon, const void* images, void* grad, void* out2, void* workspace,
                 size_t workspace_bytes, size_t n, float w1, float w2, dmvae_stream_t stream);

/* One LPIPS level (utils/lpips.py:86-94): out[0] (+)= mean_{n,hw} sum_c w_c (f0/(|f0|+1e-10) -
 * f1/(|f1|+1e-10))^2 for NHWC bf16 features [n][hw][c].  df1 (optional bf16) = gscale * d(sum)/d f1. */
int dmvae_lpips_diff(const void* f0, const void* f1, const void* lin_w, void* df1, void* out, void* workspace,
                     size_t workspace_bytes, int n, int hw, int c, float gscale, int accumulate,
                     dmvae_stream_t stream);
/* dmvae_lpips_diff plus the 2x2 max pool that follows the tapped level in the VGG16 trunk (utils/lpips.py:126-135), of BOTH branches, from the one read of the
 * features: pool0 / pool1 = the pooled f0 / f1, bf16 [n][h/2][w/2][c] each (the two halves of the trunk's 2n-image tensor); h, w even.  Same df1 and pooled
 * values as the two separate calls; the level value differs by the order of its f32 partial sums only. */
int dmvae_lpips_diff_pool(const void* f0, const void* f1, const void* lin_w, void* df1, void* out, void* pool0, void* pool1, void* workspace,
                          size_t workspace_bytes, int n, int h, int w, int c, float gscale, int accumulate, dmvae_stream_t stream);

/* DMD score-gradient loss (train_dmd.py:204-230; toy_example_2d/dmd.py:349-360), f32 [batch][per_sample]:
 * pre : xt = t*x1 + (1-t)*x0                                       (ICPlan, path.py:114-136)
 * post: CFG combine, pred = xt + v*(1-t), grad = (p_real-p_student)/mean|p_real| (weight_factor!=0),
 *       nan_to_num; out2 = { 0.5*mean(grad^2), mean_b ||grad_b|| }; dlatents = grad/numel. */
int dmvae_dmd_pre(const void* x1, const void* x0, const void* t, void* xt, int batch, int per_sample,
                  dmvae_stream_t stream);
int dmvae_dmd_post(const void* x1, const void* xt, const void* t, const void* v_teacher, const void* v_teacher_u,
                   const void* v_student, const void* v_student_u, void* dlatents, void* out2, void* workspace,
                   size_t workspace_bytes, int batch, int per_sample, float cfg, int weight_factor,
                   dmvae_stream_t stream);

/* Build-defined (no reference counterpart): per-latent KL of batch moments vs N(0,1) and batched
 * RBF-mixture MMD^2 between z[g] ([n][32] f32) and y[g] ([m][32] f32), fused in one pass over z.
 * kl: [33] f32 (32 channels + their mean); mmd: [groups] f32;
 * dz (optional): w_kl * d mean(kl)/dz + w_mmd * d mean(mmd)/dz.  Any n, m > 0 (128-row tiles, 256-column LDS chunks);
 * workspace >= dmvae_kl_mmd_workspace(groups, n, m) bytes.  m = 0 (y, mmd may be NULL) runs the KL moment pass and its
 * gradient alone.  Deterministic (fixed-order reductions). */
size_t dmvae_kl_mmd_workspace(int groups, int n, int m);
int dmvae_kl_mmd(const void* z, const void* y, void* kl, void* mmd, void* dz, void* workspace,
                 size_t workspace_bytes, int groups, int n, int m, int d, float w_kl, float w_mmd,
                 dmvae_stream_t stream);

/* Reparameterised sample + posterior-form KL of a diagonal-Gaussian latent head.  BUILD-DEFINED, parity unpinned: the reference's VAE.forward is
 * deterministic (models/vae.py:90-98, no mean / log-variance split); BASELINE.json's north_star names the hook ("encoder -> reparameterise -> decoder",
 * "per-latent KL"), SURVEY.md 0 asks for it OFF by default, 8a row a15 gives the posterior form.  Host caller: models/vae.py VAE(reparameterize=True).
 * moments [rows][2C] = (mu | logvar) per row, f32 or bf16 (bf16_io); eps [rows][C] f32 caller-drawn N(0,1), NULL = the posterior mode (z = mu);
 * z [rows][C] (type of moments; may be NULL) = mu + exp(logvar/2)*eps; kl [C+1] f32: kl[c] = mean_r 0.5*(mu^2 + exp(lv) - 1 - lv), kl[C] = mean_c.
 * C a power of two in [4, 256]; workspace >= dmvae_reparam_kl_workspace(rows, C).  Deterministic (fixed-order reductions). */
size_t dmvae_reparam_kl_workspace(size_t rows, int C);
int dmvae_reparam_kl_fwd(const void* moments, const void* eps, void* z, void* kl, void* workspace, size_t workspace_bytes,
                         size_t rows, int C, int bf16_io, dmvae_stream_t stream);
/* d moments [rows][2C] = (dz + g*mu/(rows*C) | dz*0.5*exp(lv/2)*eps + g*0.5*(exp(lv)-1)/(rows*C)), g = w_kl * (g_kl ? *g_kl : 1) with g_kl a DEVICE
 * pointer to one f32 (the upstream gradient of kl[C]); dz [rows][C] (type of moments) may be NULL (KL term only).  Same build-defined status as above. */
int dmvae_reparam_kl_bwd(const void* moments, const void* eps, const void* dz, const void* g_kl, float w_kl, void* dmoments,
                         size_t rows, int C, int bf16_io, dmvae_stream_t stream);

/* ---- optimiser tail on flat f32 buffers (train_tokenizer.py:140-150,382,415-419) ---------------- */

/* norm_out3 = { ||g||_2, min(1, max_norm/(norm+1e-6)), sum of squares }; accumulate_prev!=0 adds the
 * previous call's sum of squares (norm over several buffers).  workspace >= 8 KiB.  Reference: clip_grad_norm_, train_tokenizer.py:382 (train_dmd.py:545,573). */
int dmvae_grad_norm(const void* grads, void* norm_out3, void* workspace, size_t workspace_bytes, size_t n,
                    float max_norm, int accumulate_prev, dmvae_stream_t stream);
/* p,m,v (and ema when non-NULL) updated in place: g*=clip (norm_out3[1], NULL = no clip); decoupled
 * weight decay; bias-corrected Adam (torch.optim.AdamW semantics); ema = ema*decay + p*(1-decay).  Reference: torch.optim.AdamW.step + update_ema, train_tokenizer.py:140-150,415-419. */
int dmvae_adamw_ema_step(void* params, const void* grads, void* exp_avg, void* exp_avg_sq, void* ema,
                         const void* norm_out3, size_t n, float lr, float beta1, float beta2, float eps,
                         float weight_decay, int step, float ema_decay, dmvae_stream_t stream);
/* Same step, also writing bf16_shadow[i] = bf16(params[i]) (round to nearest even): the copy torch.autocast(bfloat16) makes of every Linear weight
 * on every forward (`weight.to(bfloat16)`), produced once per optimiser step in the pass that already holds the new value.  Reference: torch.optim.AdamW.step + update_ema, train_tokenizer.py:140-150,415-419. */
int dmvae_adamw_ema_step_shadow(void* params, const void* grads, void* exp_avg, void* exp_avg_sq, void* ema, void* bf16_shadow,
                                const void* norm_out3, size_t n, float lr, float beta1, float beta2, float eps,
                                float weight_decay, int step, float ema_decay, dmvae_stream_t stream);

/* ---- downstream consumers: SDE sampler state update and image -> uint8 (sample_50k.py:142-164) ---- */

/* One Euler-Maruyama step of diffusion/transport/integrators.py:27-35 on the whole state with the drift of transport.py:254-257 and
 * the velocity -> score conversion of path.py:74-89 folded in; every coefficient depends on t only (shared by the batch) and is
 * passed as the f32 scalar the reference's graph holds:
 *   score = (rar * v - x) / var;  drift = v + diff * score;  mean = x + drift * dt;  x' = mean + sqrt_2diff * (w * sqrt_dt)
 * x, w, x_out, mean_out: [n] f32; v: [n] bf16 (v_is_bf16 != 0, the autocast model output) or f32.  w = NULL: x_out = mean (the
 * "Mean" / "Euler" last step, transport.py:275-295, with dt = last_step_size).  x_out or mean_out may be NULL.  n % 4 == 0.
 * Operation order and rounding are the reference's (no FMA contraction): bit-identical to the PyTorch-CPU f32 result for the same v. */
int dmvae_sde_euler_step(const void* x, const void* v, int v_is_bf16, const void* w, void* x_out, void* mean_out, size_t n,
                         float rar, float var, float diff, float dt, float sqrt_2diff, float sqrt_dt, dmvae_stream_t stream);
/* out[npix][c] uint8 = (uint8) clamp(127.5 * y + 128, 0, 255) for y [npix][c_stride] f32 (NHWC decoder output, first c channels):
 * sample_50k.py:151 without the NCHW round trip.  round_bf16 != 0 rounds y to bf16 first (an autocast decoder's `.float()`). */
int dmvae_image_to_u8(const void* y, void* out, size_t npix, int c, int c_stride, int round_bf16, dmvae_stream_t stream);

/* ---- fp32 parity mode (DMVAE_PARITY=1; csrc/parity.hip) ---------------------------------------------------------------------------
 * north_star: "match the reference PyTorch-CPU path within 1e-4 relative fp32".  In this mode activations are f32 NHWC and every
 * contraction still runs on the bf16 MFMA kernels above: dmvae_split3_bf16 splits an f32 operand EXACTLY into hi + mid + lo bf16 terms and
 * lays the six partial products of order <= 2 along the reduction dimension, so one launch of dmvae_conv2d_nhwc_fwd / _wgrad /
 * dmvae_gemm_nt_batched / dmvae_gemm_tn_batched with a 6x longer reduction accumulates x*w to ~2^-24 in its f32 accumulator.  The other
 * entry points are the f32-in / f32-out forms of the HBM-bound kernels (f64 statistics), replacing the same reference sites as their bf16
 * counterparts (models/flux_ae.py:21-107,239-269; models/vae.py:56-65; utils/lpips.py:86-162). */

/* x [rows][cols] f32 -> bf16 parts; part q of element (r, c) is written to
 *   out[(r / rows_per_batch) * batch_stride + q * part_stride + (r % rows_per_batch) * row_stride + c],  q = 0..5
 * pattern 0 (activation side): [hi, mid, lo, hi, mid, hi];  pattern 1 (weight side): [hi, hi, hi, mid, mid, lo].  Reference: Parity mode only: operands of the f32-emulating products for models/flux_ae.py:21-107,239-269. */
int dmvae_split3_bf16(const void* x, void* out, size_t rows, int cols, size_t rows_per_batch, size_t batch_stride, size_t part_stride,
                      size_t row_stride, int pattern, dmvae_stream_t stream);
/* GroupNorm on f32 NHWC (statistics accumulated in f64); act as dmvae_groupnorm_apply.  nn.GroupNorm at flux_ae.py:28,62,64,236. */
int dmvae_groupnorm_stats_f32(const void* x, void* stats, int n, int hw, int c, int groups, float eps, dmvae_stream_t stream);
int dmvae_groupnorm_apply_f32(const void* x, const void* stats, const void* gamma, const void* beta, void* y, int n, int hw, int c,
                              int groups, int act, dmvae_stream_t stream);
size_t dmvae_groupnorm_f32_workspace(int n, int c, int groups);
/* dx = GroupNorm backward of da (+ dres), dgamma / dbeta (NULL to skip); inv_count <= 0 selects 1 / (hw * c / groups).  Reference: Parity mode: autograd of Normalize(), models/flux_ae.py:28. */
int dmvae_groupnorm_bwd_f32(const void* da, const void* x, const void* dres, const void* stats, const void* gamma, const void* beta, void* dx,
                            void* dgamma, void* dbeta, void* workspace, size_t workspace_bytes, int n, int hw, int c, int groups, int act,
                            int accumulate, float inv_count, dmvae_stream_t stream);
/* Elementwise f32 family.  op 0: out = act(a + b) (b may be NULL; act 0 none, 1 SiLU, 2 ReLU, 4 LeakyReLU(param), 3: out = b > 0 ? a : 0);
 * 1: SiLU(a); 2: b * SiLU'(a); 3: a * (b > 0 ? 1 : param) (ReLU / LeakyReLU backward from the saved output b); 4: GELU(a) (erf form);
 * 5: b * GELU'(a); 6: a + b * g[i % cols] (LayerScale + residual); 7: a * param.  Reference: Parity mode: swish models/flux_ae.py:24, residual adds :75,82, nn.ReLU utils/lpips.py:116-153, LeakyReLU models/patchgan.py:125-147. */
int dmvae_eltwise_f32(int op, const void* a, const void* b, const void* g, void* out, size_t n, int cols, int act, float param,
                      dmvae_stream_t stream);
/* Row softmax with f32 probabilities (flux_ae.py:47) and its backward dS = scale * P .* (dP - sum(dP .* P)). */
int dmvae_softmax_rows_fwd_f32(const void* s, void* p, int rows, int cols, float scale, dmvae_stream_t stream);
int dmvae_softmax_rows_bwd_f32(const void* dp, const void* p, void* ds, int rows, int cols, float scale, dmvae_stream_t stream);
/* 2x2 pools on f32 NHWC; n, h, w = POOLED size.  op 0: sum (backward of nearest x2, flux_ae.py:104); 1: max (lpips.py VGG trunk);
 * 2: out [n,2h,2w,c] = ReLU-masked max-pool backward of a (may be NULL) at the argmax of x's window, plus extra (may be NULL).  Reference: models/flux_ae.py:104;
 * utils/lpips.py:126-135. */
int dmvae_pool2x2_f32(int op, const void* a, const void* x, const void* extra, void* out, int n, int h, int w, int c, dmvae_stream_t stream);
/* Parity mode: the NCHW f32 image / latent as an f32 NHWC operand (channels zero-padded to c_pad), models/vae.py:56-65. */
int dmvae_nchw_f32_to_nhwc_f32(const void* src, void* dst, int n, int c, int hw, int c_pad, dmvae_stream_t stream);
/* dmvae_lpips_diff on f32 features; workspace >= 2048 * 8 bytes.  Reference: Parity mode: utils/lpips.py:86-94,107-113,156-162. */
int dmvae_lpips_diff_f32(const void* f0, const void* f1, const void* lin_w, void* df1, void* out, void* workspace, size_t workspace_bytes, int n,
                         int hw, int c, float gscale, int accumulate, dmvae_stream_t stream);
/* LayerNorm over the last dimension, f32 in / f32 out (timm ViT block reached through models/vae.py:47-53). */
int dmvae_layernorm_f32(const void* x, const void* gamma, const void* beta, void* y, int rows, int cols, float eps, dmvae_stream_t stream);

/* ---- fp32 parity mode, transformer rows (csrc/parity_dit.hip): the elementwise / normalisation steps of LightningDiT and of the trainable ViT block, forward and
 * backward, f32 in / f32 out with f64 row statistics -- with them (and the split-operand GEMMs) dmvae_amd/models/lightningdit_parity.py reaches 1e-4 against the
 * reference's f32 captures.  A verification mode, not a training mode.  `shift` / `scale` / `g` are chunks of an adaLN output [B][ld_mod] (pointer at the chunk). */

/* y = x * rstd * w * (1 + scale[b]) + shift[b], rstd [rows] kept (scale / shift may be NULL).  Reference: rms_norm.py:52-76 + modulate, lightningdit.py:27-31,241-250. */
int dmvae_rms_modulate_fwd_f32(const void* x, const void* w, const void* shift, const void* scale, void* y, void* rstd, size_t rows, int c,
                               int rows_per_sample, int ld_mod, float eps, dmvae_stream_t stream);
/* dx; gw = dy * xh * (1 + scale) (column sum = d w); gs = dy * xh * w (per-sample column sum = d scale; with scale).  Reference: the backward of rms_norm.py:52-76 / lightningdit.py:27-31. */
int dmvae_rms_modulate_bwd_f32(const void* dy, const void* x, const void* w, const void* scale, const void* rstd, void* dx, void* gw, void* gs,
                               size_t rows, int c, int rows_per_sample, int ld_mod, dmvae_stream_t stream);
/* LayerNorm backward: dx, gw = dy * xh (column sum = d gamma; d beta = column sum of dy).  Reference: timm Block's norm1 / norm2 through models/vae.py:47-53, trained at train_dmd.py:518-520. */
int dmvae_layernorm_bwd_full_f32(const void* dy, const void* x, const void* gamma, void* dx, void* gw, size_t rows, int c, float eps, dmvae_stream_t stream);
/* op 0: out = a + g[b] * b (gated residual, lightningdit.py:245,249)   op 1: out = g[b] * a   op 2: out = a * b. */
int dmvae_bcast_rows_f32(int op, const void* a, const void* b, const void* g, void* out, size_t rows, int c, int rows_per_sample, int ld_mod,
                         dmvae_stream_t stream);
/* out [groups][c] (+)= sum over the rows of x [groups][rows][c], f64, row order: the per-sample / per-parameter sums of the backward (lightningdit.py:241-250 adaLN chunks, norm weights). */
int dmvae_colsum_groups_f32(const void* x, void* out, int groups, int rows, int c, int accumulate, dmvae_stream_t stream);
/* g = silu(x1) * x2 over [rows][2 hidden] = [x1 | x2], and its backward.  Reference: swiglu_ffn.py:31-36. */
int dmvae_swiglu_fwd_f32(const void* x12, void* g, size_t rows, int hidden, dmvae_stream_t stream);
int dmvae_swiglu_bwd_f32(const void* dg, const void* x12, void* dx12, size_t rows, int hidden, dmvae_stream_t stream);
/* QK RMSNorm + RoPE: qkv [B][N][3][H][D] -> q, k, v [B*H][N][d_pad] (zero columns past D), rstd [2][B*N*H]; backward -> dqkv, gwq / gwk [B*N*H][D] (column sums = d w).
 * Reference: lightningdit.py:66-88 (q_norm / k_norm, rope), pos_embed.py:37-41,135. */
int dmvae_qknorm_rope_fwd_f32(const void* qkv, const void* wq, const void* wk, const void* cosb, const void* sinb, void* q, void* k, void* v, void* rstd,
                              int batch, int tokens, int heads, int d, int d_pad, float eps, dmvae_stream_t stream);
int dmvae_qknorm_rope_bwd_f32(const void* dq, const void* dk, const void* dv, const void* qkv, const void* wq, const void* wk, const void* cosb,
                              const void* sinb, const void* rstd, void* dqkv, void* gwq, void* gwk, int batch, int tokens, int heads, int d, int d_pad,
                              dmvae_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DMVAE_HIP_H */
