// fp32 parity mode, transformer rows (round 6): the elementwise / normalisation steps of LightningDiT and of the trainable ViT block -- forward AND backward -- as
// f32-in / f32-out kernels with f64 row statistics, so that these models reach north_star's 1e-4 against the reference's f32 captures the way the decoder does
// (dmvae_amd/parity.py: every contraction stays on the production MFMA GEMMs over exactly split bf16 operands; what is here replaces the bf16 elementwise
// kernels of csrc/dit.hip / vit_bwd.hip, which round to bf16 where autocast does).  A verification mode: one workgroup or wave per row, no tuning.
//
// Reference sites: diffusion/lightningdit/rms_norm.py:52-76 (RMSNorm), lightningdit.py:27-31 (modulate), :66-88 (QK-norm + RoPE: pos_embed.py:37-41,135),
// :241-250 (gated residual), swiglu_ffn.py:31-36; timm Block through models/vae.py:47-53 (LayerNorm backward, train_dmd.py:518-520).
#include "common.h"
#include "dmvae_hip.h"

namespace dmvae_parity_dit {

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double block_sum_d(double v, double* sh) {   // 256 threads; fixed order
  v = wave_sum_d(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__device__ __forceinline__ float sigmoid_exact(float x) { return 1.0f / (1.0f + expf(-x)); }

// y = x * rstd * w * (1 + scale[b]) + shift[b]; rstd[row] kept for the backward.  shift / scale: chunks of the adaLN output, [B][ldm] with the chunk's offset
// already added to the pointer; either may be null.  w null: no weight (the LayerNorm-free case is not needed).
__global__ __launch_bounds__(256) void rms_mod_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ shift,
                                                          const float* __restrict__ scale, float* __restrict__ y, float* __restrict__ rstd, int C, int rps, int ldm,
                                                          float eps) {
  __shared__ double sh[4];
  const size_t r = blockIdx.x;
  const int b = (int)(r / rps);
  const float* xr = x + r * C;
  double s = 0.0;
  for (int c = threadIdx.x; c < C; c += 256) { const double v = xr[c]; s += v * v; }
  s = block_sum_d(s, sh);
  const float rs = (float)(1.0 / sqrt(s / C + (double)eps));
  if (threadIdx.x == 0) rstd[r] = rs;
  for (int c = threadIdx.x; c < C; c += 256) {
    float v = xr[c] * rs * w[c];
    if (scale) v *= 1.0f + scale[(size_t)b * ldm + c];
    if (shift) v += shift[(size_t)b * ldm + c];
    y[r * C + c] = v;
  }
}

// dx = rstd * (a - xh * mean(a * xh)),  a = dy * w * (1 + scale), xh = x * rstd;  gw = dy * xh * (1 + scale) (its column sum over all rows is d w),
// gs = dy * xh * w (its per-sample column sum is d scale; written when scale != null).  d shift is the per-sample column sum of dy itself.
__global__ __launch_bounds__(256) void rms_mod_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ scale, const float* __restrict__ rstd, float* __restrict__ dx,
                                                          float* __restrict__ gw, float* __restrict__ gs, int C, int rps, int ldm) {
  __shared__ double sh[4];
  const size_t r = blockIdx.x;
  const int b = (int)(r / rps);
  const float rs = rstd[r];
  const float* xr = x + r * C;
  const float* dr = dy + r * C;
  double dot = 0.0;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float m1 = scale ? 1.0f + scale[(size_t)b * ldm + c] : 1.0f;
    dot += (double)(dr[c] * w[c] * m1) * (double)(xr[c] * rs);
  }
  dot = block_sum_d(dot, sh) / C;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float m1 = scale ? 1.0f + scale[(size_t)b * ldm + c] : 1.0f;
    const float xh = xr[c] * rs, a = dr[c] * w[c] * m1;
    dx[r * C + c] = rs * (a - xh * (float)dot);
    gw[r * C + c] = dr[c] * xh * m1;
    if (gs) gs[r * C + c] = dr[c] * xh * w[c];
  }
}

// LayerNorm backward over the last dim: dx = rstd * (a - mean(a) - xh * mean(a * xh)), a = dy * gamma, xh = (x - mu) * rstd (recomputed in f64);
// gw = dy * xh (column sum = d gamma); d beta is the column sum of dy.
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
                                                            float* __restrict__ dx, float* __restrict__ gw, int C, float eps) {
  __shared__ double sh[4];
  const size_t r = blockIdx.x;
  const float* xr = x + r * C;
  const float* dr = dy + r * C;
  double s = 0.0, ss = 0.0;
  for (int c = threadIdx.x; c < C; c += 256) { const double v = xr[c]; s += v; ss += v * v; }
  s = block_sum_d(s, sh);
  ss = block_sum_d(ss, sh);
  const double mean = s / C;
  double var = ss / C - mean * mean;
  if (var < 0) var = 0;
  const float mu = (float)mean, rs = (float)(1.0 / sqrt(var + (double)eps));
  double sa = 0.0, sax = 0.0;
  for (int c = threadIdx.x; c < C; c += 256) {
    const double a = (double)(dr[c] * gamma[c]), xh = (double)((xr[c] - mu) * rs);
    sa += a; sax += a * xh;
  }
  sa = block_sum_d(sa, sh) / C;
  sax = block_sum_d(sax, sh) / C;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float xh = (xr[c] - mu) * rs, a = dr[c] * gamma[c];
    dx[r * C + c] = rs * (a - (float)sa - xh * (float)sax);
    gw[r * C + c] = dr[c] * xh;
  }
}

// per-sample broadcast steps over [R][C] with g [B][ldm] (an adaLN chunk):  op 0: out = a + g[b] * b_ (gated residual)   op 1: out = g[b] * a   op 2: out = a * b_
__global__ void bcast_kernel(int op, const float* __restrict__ a, const float* __restrict__ b_, const float* __restrict__ g, float* __restrict__ out, size_t n,
                             int C, int rps, int ldm) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t r = i / C;
    const int c = (int)(i - r * C);
    const float gv = g ? g[(size_t)(r / rps) * ldm + c] : 1.0f;
    out[i] = op == 0 ? a[i] + gv * b_[i] : (op == 1 ? gv * a[i] : a[i] * b_[i]);
  }
}

// out[g][c] = sum over the R rows of group g of x[g][r][c], f64, row order (deterministic).  One thread per (group, column).
__global__ void colsum_groups_kernel(const float* __restrict__ x, float* __restrict__ out, int G, int R, int C, int accumulate) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)G * C) return;
  const int g = (int)(i / C), c = (int)(i - (size_t)g * C);
  const float* p = x + ((size_t)g * R) * C + c;
  double s = 0.0;
  for (int r = 0; r < R; r++) s += p[(size_t)r * C];
  out[i] = (accumulate ? out[i] : 0.f) + (float)s;
}

// SwiGLU on [R][2H] = [x1 | x2]:  g = silu(x1) * x2;  backward: dx1 = dg * x2 * silu'(x1), dx2 = dg * silu(x1)
__global__ void swiglu_fwd_kernel(const float* __restrict__ x12, float* __restrict__ g, size_t R, int H) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < R * H; i += (size_t)gridDim.x * 256) {
    const size_t r = i / H;
    const int h = (int)(i - r * H);
    const float x1 = x12[r * 2 * H + h], x2 = x12[r * 2 * H + H + h];
    g[i] = x1 * sigmoid_exact(x1) * x2;
  }
}
__global__ void swiglu_bwd_kernel(const float* __restrict__ dg, const float* __restrict__ x12, float* __restrict__ dx12, size_t R, int H) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < R * H; i += (size_t)gridDim.x * 256) {
    const size_t r = i / H;
    const int h = (int)(i - r * H);
    const float x1 = x12[r * 2 * H + h], x2 = x12[r * 2 * H + H + h], sg = sigmoid_exact(x1), d = dg[i];
    dx12[r * 2 * H + h] = d * x2 * sg * (1.0f + x1 * (1.0f - sg));
    dx12[r * 2 * H + H + h] = d * x1 * sg;
  }
}

// QK-norm + RoPE, one wave per (sample, token, head) and per q / k:  xh = x * rstd (rstd over the head's D channels), v = xh * w, out = v * cos + rot(v) * sin with
// rot(v)[2i] = -v[2i + 1], rot(v)[2i + 1] = v[2i] (pos_embed.py:37-41).  qkv [B][N][3][H][D] -> q, k, v [B * H][N][Dp] (columns D .. Dp - 1 zero: the GEMMs'
// reduction granule); rstd [2][B * N * H].  D even, D <= 256.
__global__ __launch_bounds__(256) void qknorm_rope_fwd_kernel(const float* __restrict__ qkv, const float* __restrict__ wq, const float* __restrict__ wk,
                                                              const float* __restrict__ cosb, const float* __restrict__ sinb, float* __restrict__ q,
                                                              float* __restrict__ k, float* __restrict__ v, float* __restrict__ rstd, int B, int N, int H, int D,
                                                              int Dp, float eps) {
  const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // (b, n, h)
  const int lane = threadIdx.x & 63;
  const size_t rows = (size_t)B * N * H;
  if (row >= rows) return;
  const int h = (int)(row % H), n = (int)((row / H) % N), b = (int)(row / ((size_t)H * N));
  const size_t dst = (((size_t)b * H + h) * N + n) * Dp;
  for (int which = 0; which < 3; which++) {
    const float* src = qkv + (((size_t)b * N + n) * 3 + which) * H * D + (size_t)h * D;
    float* out = (which == 0 ? q : (which == 1 ? k : v)) + dst;
    if (which == 2) {
      for (int d = lane; d < Dp; d += 64) out[d] = d < D ? src[d] : 0.f;
      continue;
    }
    const float* w = which == 0 ? wq : wk;
    double s = 0.0;
    for (int d = lane; d < D; d += 64) { const double t = src[d]; s += t * t; }
    s = wave_sum_d(s);
    const float rs = (float)(1.0 / sqrt(s / D + (double)eps));
    if (lane == 0) rstd[(size_t)which * rows + row] = rs;
    for (int d = lane; d < Dp; d += 64) {
      float o = 0.f;
      if (d < D) {
        const int dp = d ^ 1;
        const float vd = src[d] * rs * w[d], vp = src[dp] * rs * w[dp];
        const float rot = (d & 1) ? vp : -vp;
        o = vd * cosb[(size_t)n * D + d] + rot * sinb[(size_t)n * D + d];
      }
      out[d] = o;
    }
  }
}

// backward: dq, dk, dv [B * H][N][Dp] -> dqkv [B][N][3][H][D];  gwq / gwk [B * N * H][D] = d v * xh (column sums = d w)
__global__ __launch_bounds__(256) void qknorm_rope_bwd_kernel(const float* __restrict__ dq, const float* __restrict__ dk, const float* __restrict__ dv,
                                                              const float* __restrict__ qkv, const float* __restrict__ wq, const float* __restrict__ wk,
                                                              const float* __restrict__ cosb, const float* __restrict__ sinb, const float* __restrict__ rstd,
                                                              float* __restrict__ dqkv, float* __restrict__ gwq, float* __restrict__ gwk, int B, int N, int H, int D,
                                                              int Dp) {
  const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const size_t rows = (size_t)B * N * H;
  if (row >= rows) return;
  const int h = (int)(row % H), n = (int)((row / H) % N), b = (int)(row / ((size_t)H * N));
  const size_t srcg = (((size_t)b * H + h) * N + n) * Dp;
  for (int which = 0; which < 3; which++) {
    const size_t off = (((size_t)b * N + n) * 3 + which) * H * D + (size_t)h * D;
    const float* g = (which == 0 ? dq : (which == 1 ? dk : dv)) + srcg;
    if (which == 2) {
      for (int d = lane; d < D; d += 64) dqkv[off + d] = g[d];
      continue;
    }
    const float* w = which == 0 ? wq : wk;
    float* gwo = (which == 0 ? gwq : gwk) + row * D;
    const float rs = rstd[(size_t)which * rows + row];
    const float* x = qkv + off;
    // d v (gradient at the weighted, un-rotated vector): transpose of the rotation
    double dot = 0.0;
    for (int d = lane; d < D; d += 64) {
      const int dp = d ^ 1;
      // out[d] = v[d] cos[d] + (d odd ? v[dp] : -v[dp]) sin[d]  =>  d v[d] = g[d] cos[d] + (dp odd ? +1 : -1) g[dp] sin[dp]   (v[d] enters out[dp] with the sign of dp's rule)
      const float dvd = g[d] * cosb[(size_t)n * D + d] + ((dp & 1) ? g[dp] : -g[dp]) * sinb[(size_t)n * D + dp];
      const float xh = x[d] * rs;
      dot += (double)(dvd * w[d]) * (double)xh;
    }
    dot = wave_sum_d(dot) / D;
    for (int d = lane; d < D; d += 64) {
      const int dp = d ^ 1;
      const float dvd = g[d] * cosb[(size_t)n * D + d] + ((dp & 1) ? g[dp] : -g[dp]) * sinb[(size_t)n * D + dp];
      const float xh = x[d] * rs;
      dqkv[off + d] = rs * (dvd * w[d] - xh * (float)dot);
      gwo[d] = dvd * xh;
    }
  }
}

static inline unsigned grid_for(size_t n) { const size_t g = (n + 255) / 256; return (unsigned)(g > 65535 ? 65535 : (g ? g : 1)); }

}  // namespace dmvae_parity_dit
using namespace dmvae_parity_dit;

extern "C" int dmvae_rms_modulate_fwd_f32(const void* x, const void* w, const void* shift, const void* scale, void* y, void* rstd, size_t rows, int c,
                                          int rows_per_sample, int ld_mod, float eps, hipStream_t stream) {
  DMVAE_CHECK_ARG(x && w && y && rstd && rows > 0 && c > 0 && rows_per_sample > 0 && rows < (1u << 31), "rms_modulate_fwd_f32: bad arguments");
  hipLaunchKernelGGL(rms_mod_fwd_kernel, dim3((unsigned)rows), dim3(256), 0, stream, (const float*)x, (const float*)w, (const float*)shift, (const float*)scale,
                     (float*)y, (float*)rstd, c, rows_per_sample, ld_mod, eps);
  DMVAE_CHECK_LAUNCH();
  return 0;
}
extern "C" int dmvae_rms_modulate_bwd_f32(const void* dy, const void* x, const void* w, const void* scale, const void* rstd, void* dx, void* gw, void* gs,
                                          size_t rows, int c, int rows_per_sample, int ld_mod, hipStream_t stream) {
  DMVAE_CHECK_ARG(dy && x && w && rstd && dx && gw && rows > 0 && c > 0 && rows_per_sample > 0 && rows < (1u << 31), "rms_modulate_bwd_f32: bad arguments");
  DMVAE_CHECK_ARG(!scale == !gs, "rms_modulate_bwd_f32: gs goes with scale");
  hipLaunchKernelGGL(rms_mod_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, stream, (const float*)dy, (const float*)x, (const float*)w, (const float*)scale,
                     (const float*)rstd, (float*)dx, (float*)gw, (float*)gs, c, rows_per_sample, ld_mod);
  DMVAE_CHECK_LAUNCH();
  return 0;
}
extern "C" int dmvae_layernorm_bwd_full_f32(const void* dy, const void* x, const void* gamma, void* dx, void* gw, size_t rows, int c, float eps, hipStream_t stream) {
  DMVAE_CHECK_ARG(dy && x && gamma && dx && gw && rows > 0 && c > 0 && rows < (1u << 31), "layernorm_bwd_full_f32: bad arguments");
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, stream, (const float*)dy, (const float*)x, (const float*)gamma, (float*)dx,
                     (float*)gw, c, eps);
  DMVAE_CHECK_LAUNCH();
  return 0;
}
extern "C" int dmvae_bcast_rows_f32(int op, const void* a, const void* b, const void* g, void* out, size_t rows, int c, int rows_per_sample, int ld_mod,
                                    hipStream_t stream) {
  DMVAE_CHECK_ARG(a && out && rows > 0 && c > 0 && op >= 0 && op <= 2 && rows_per_sample > 0, "bcast_rows_f32: bad arguments");
  DMVAE_CHECK_ARG((op == 1 || b) && (op == 2 || g), "bcast_rows_f32: op %d misses an operand", op);
  hipLaunchKernelGGL(bcast_kernel, dim3(grid_for(rows * c)), dim3(256), 0, stream, op, (const float*)a, (const float*)b, (const float*)g, (float*)out, rows * c, c,
                     rows_per_sample, ld_mod);
  DMVAE_CHECK_LAUNCH();
  return 0;
}
extern "C" int dmvae_colsum_groups_f32(const void* x, void* out, int groups, int rows, int c, int accumulate, hipStream_t stream) {
  DMVAE_CHECK_ARG(x && out && groups > 0 && rows > 0 && c > 0, "colsum_groups_f32: bad arguments");
  hipLaunchKernelGGL(colsum_groups_kernel, dim3((unsigned)(((size_t)groups * c + 255) / 256)), dim3(256), 0, stream, (const float*)x, (float*)out, groups, rows, c,
                     accumulate);
  DMVAE_CHECK_LAUNCH();
  return 0;
}
extern "C" int dmvae_swiglu_fwd_f32(const void* x12, void* g, size_t rows, int hidden, hipStream_t stream) {
  DMVAE_CHECK_ARG(x12 && g && rows > 0 && hidden > 0, "swiglu_fwd_f32: bad arguments");
  hipLaunchKernelGGL(swiglu_fwd_kernel, dim3(grid_for(rows * hidden)), dim3(256), 0, stream, (const float*)x12, (float*)g, rows, hidden);
  DMVAE_CHECK_LAUNCH();
  return 0;
}
extern "C" int dmvae_swiglu_bwd_f32(const void* dg, const void* x12, void* dx12, size_t rows, int hidden, hipStream_t stream) {
  DMVAE_CHECK_ARG(dg && x12 && dx12 && rows > 0 && hidden > 0, "swiglu_bwd_f32: bad arguments");
  hipLaunchKernelGGL(swiglu_bwd_kernel, dim3(grid_for(rows * hidden)), dim3(256), 0, stream, (const float*)dg, (const float*)x12, (float*)dx12, rows, hidden);
  DMVAE_CHECK_LAUNCH();
  return 0;
}
extern "C" int dmvae_qknorm_rope_fwd_f32(const void* qkv, const void* wq, const void* wk, const void* cosb, const void* sinb, void* q, void* k, void* v, void* rstd,
                                         int batch, int tokens, int heads, int d, int d_pad, float eps, hipStream_t stream) {
  DMVAE_CHECK_ARG(qkv && wq && wk && cosb && sinb && q && k && v && rstd, "qknorm_rope_fwd_f32: null pointer");
  DMVAE_CHECK_ARG(batch > 0 && tokens > 0 && heads > 0 && d > 0 && d % 2 == 0 && d_pad >= d, "qknorm_rope_fwd_f32: need an even head dim <= d_pad");
  const size_t rows = (size_t)batch * tokens * heads;
  hipLaunchKernelGGL(qknorm_rope_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, (const float*)qkv, (const float*)wq, (const float*)wk,
                     (const float*)cosb, (const float*)sinb, (float*)q, (float*)k, (float*)v, (float*)rstd, batch, tokens, heads, d, d_pad, eps);
  DMVAE_CHECK_LAUNCH();
  return 0;
}
extern "C" int dmvae_qknorm_rope_bwd_f32(const void* dq, const void* dk, const void* dv, const void* qkv, const void* wq, const void* wk, const void* cosb,
                                         const void* sinb, const void* rstd, void* dqkv, void* gwq, void* gwk, int batch, int tokens, int heads, int d, int d_pad,
                                         hipStream_t stream) {
  DMVAE_CHECK_ARG(dq && dk && dv && qkv && wq && wk && cosb && sinb && rstd && dqkv && gwq && gwk, "qknorm_rope_bwd_f32: null pointer");
  DMVAE_CHECK_ARG(batch > 0 && tokens > 0 && heads > 0 && d > 0 && d % 2 == 0 && d_pad >= d, "qknorm_rope_bwd_f32: need an even head dim <= d_pad");
  const size_t rows = (size_t)batch * tokens * heads;
  hipLaunchKernelGGL(qknorm_rope_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, (const float*)dq, (const float*)dk, (const float*)dv,
                     (const float*)qkv, (const float*)wq, (const float*)wk, (const float*)cosb, (const float*)sinb, (const float*)rstd, (float*)dqkv, (float*)gwq,
                     (float*)gwk, batch, tokens, heads, d, d_pad);
  DMVAE_CHECK_LAUNCH();
  return 0;
}
