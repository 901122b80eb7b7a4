// fp32 parity mode (DMVAE_PARITY=1): f32 activations end to end, every contraction still on the bf16 MFMA kernels.
//
// north_star asks for agreement with the reference's PyTorch-CPU fp32 path within 1e-4.  A bf16 operand carries 2^-9 of rounding
// error, so the production kernels can only meet that bar against an oracle fed the same rounded operands.  This mode removes the
// operand rounding instead of replacing the kernels: an f32 value v is split EXACTLY into three bf16 terms
//     v = hi + mid + lo,   hi = bf16(v), mid = bf16(v - hi), lo = bf16(v - hi - mid)          (8 + 8 + 8 mantissa bits)
// and a product x*w is evaluated as the six partial products of order <= 2
//     xh*wh + xm*wh + xl*wh + xh*wm + xm*wm + xh*wl                                            (dropped terms <= 2^-24 |x w|)
// which are exact in the MFMA's f32 accumulator.  The six terms are laid out along the contraction's REDUCTION dimension (channels for
// conv forward / input gradient / NT GEMM, pixels for the weight gradient / TN GEMM), so ONE launch of the same conv_pp / conv_fwd /
// wgrad_pp / wgrad kernels with a 6x longer reduction accumulates them in f32 -- the hot kernels themselves are what the parity tests run.
//   operand A (activation side): parts [hi, mid, lo, hi, mid, hi]
//   operand B (weight side)    : parts [hi, hi,  hi, mid, mid, lo]
// Everything else in this file is the f32-in / f32-out form of the path's HBM-bound elementwise and normalisation kernels, written for
// accuracy (f64 statistics, expf / erff instead of the fast intrinsics), not for speed: one thread per element.
// Reference sites: models/flux_ae.py:21-107,239-269 (GroupNorm, swish, residual adds, attention softmax), models/vae.py:56-65 (SiLU),
// utils/lpips.py:86-94,116-162 (ReLU, max-pool, feature diff), timm ViT block (LayerNorm, GELU, LayerScale) reached through models/vae.py:47-53.
#include "common.h"
#include "dmvae_hip.h"

namespace dmvae_parity {

static inline int grid_for(size_t n, int block = 256, int cap = 8192) {
  size_t g = (n + block - 1) / block;
  return (int)(g > (size_t)cap ? cap : (g < 1 ? 1 : g));
}
#define GSTRIDE(i, total) for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (total); i += (size_t)gridDim.x * blockDim.x)

__device__ __forceinline__ double block_sum_d(double v, double* sh) {      // 256 threads
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  const double r = sh[0];
  __syncthreads();
  return r;
}
__device__ __forceinline__ float sigmoid_exact(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---- exact three-way bf16 split ------------------------------------------------------------------------------------------------------
// x [rows][cols] f32 -> six bf16 parts; element (part q, row r, col c) goes to out[(r / rpb) * batch_stride + q * part_stride + (r % rpb) * row_stride + c]
__global__ void split3_kernel(const float* __restrict__ x, bf16* __restrict__ out, size_t total, int cols, size_t rpb, size_t batch_stride,
                              size_t part_stride, size_t row_stride, int pattern) {
  GSTRIDE(i, total) {
    const size_t r = i / cols;
    const int c = (int)(i % cols);
    const float v = x[i];
    const bf16 h = (bf16)v;
    const float r1 = v - (float)h;        // exact
    const bf16 m = (bf16)r1;
    const float r2 = r1 - (float)m;       // exact
    const bf16 l = (bf16)r2;
    bf16* o = out + (r / rpb) * batch_stride + (r % rpb) * row_stride + c;
    if (pattern == 0) { o[0] = h; o[part_stride] = m; o[2 * part_stride] = l; o[3 * part_stride] = h; o[4 * part_stride] = m; o[5 * part_stride] = h; }
    else              { o[0] = h; o[part_stride] = h; o[2 * part_stride] = h; o[3 * part_stride] = m; o[4 * part_stride] = m; o[5 * part_stride] = l; }
  }
}

// ---- GroupNorm, f32 in / f32 out, f64 statistics --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, float* __restrict__ stats, int HW, int C, int G, float eps) {
  __shared__ double sh[256];
  const int n = blockIdx.y, g = blockIdx.x, cpg = C / G;
  const size_t cnt = (size_t)HW * cpg;
  const float* base = x + (size_t)n * HW * C + (size_t)g * cpg;
  double s = 0.0, ss = 0.0;
  for (size_t i = threadIdx.x; i < cnt; i += 256) {
    const double v = base[(i / cpg) * C + (i % cpg)];
    s += v; ss += v * v;
  }
  s = block_sum_d(s, sh);
  ss = block_sum_d(ss, sh);
  if (threadIdx.x == 0) {
    const double mean = s / (double)cnt;
    double var = ss / (double)cnt - mean * mean;
    if (var < 0) var = 0;
    stats[((size_t)n * G + g) * 2] = (float)mean;
    stats[((size_t)n * G + g) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}
__device__ __forceinline__ float act_fwd(float t, int act) {
  return act == 1 ? t * sigmoid_exact(t) : (act == 2 ? (t > 0.f ? t : 0.2f * t) : t);
}
__device__ __forceinline__ float act_grad(float t, int act) {       // d act(t) / dt
  if (act == 1) { const float sg = sigmoid_exact(t); return sg * (1.f + t * (1.f - sg)); }
  if (act == 2) return t > 0.f ? 1.f : 0.2f;
  return 1.f;
}
__global__ void gn_apply_kernel(const float* __restrict__ x, const float* __restrict__ stats, const float* __restrict__ gamma,
                                const float* __restrict__ beta, float* __restrict__ y, size_t total, int HW, int C, int G, int act) {
  const int cpg = C / G;
  GSTRIDE(i, total) {
    const int c = (int)(i % C);
    const size_t n = i / ((size_t)HW * C);
    const float* st = stats + (n * G + c / cpg) * 2;
    const float t = (x[i] - st[0]) * st[1] * gamma[c] + beta[c];
    y[i] = act_fwd(t, act);
  }
}
// per (image, group): AB[n][c] = (sum_p g, sum_p g * x_hat) with g = da * act'(x_hat*gamma+beta); S[n][grp] = sum_c gamma_c * AB
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(const float* __restrict__ da, const float* __restrict__ x, const float* __restrict__ stats,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ AB,
                                                            float* __restrict__ S, int HW, int C, int G, int act) {
  __shared__ double sa[256], sb[256];
  const int n = blockIdx.y, g = blockIdx.x, cpg = C / G;
  const int ci = threadIdx.x % cpg, pl = threadIdx.x / cpg, lanes = 256 / cpg;
  const int c = g * cpg + ci;
  const float mu = stats[((size_t)n * G + g) * 2], rs = stats[((size_t)n * G + g) * 2 + 1];
  const float ga = gamma[c], be = beta[c];
  double a = 0.0, b = 0.0;
  if (pl < lanes)
    for (int p = pl; p < HW; p += lanes) {
      const size_t o = ((size_t)n * HW + p) * C + c;
      const float xh = (x[o] - mu) * rs;
      const float dy = da[o] * act_grad(xh * ga + be, act);
      a += dy; b += (double)dy * xh;
    }
  sa[threadIdx.x] = pl < lanes ? a : 0.0;
  sb[threadIdx.x] = pl < lanes ? b : 0.0;
  __syncthreads();
  if ((int)threadIdx.x < cpg) {
    double ta = 0.0, tb = 0.0;
    for (int l = 0; l < lanes; l++) { ta += sa[l * cpg + threadIdx.x]; tb += sb[l * cpg + threadIdx.x]; }
    AB[((size_t)n * C + c) * 2] = (float)ta;
    AB[((size_t)n * C + c) * 2 + 1] = (float)tb;
    sa[threadIdx.x] = (double)ga * ta;
    sb[threadIdx.x] = (double)ga * tb;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < cpg; k++) { s1 += sa[k]; s2 += sb[k]; }
    S[((size_t)n * G + g) * 2] = (float)s1;
    S[((size_t)n * G + g) * 2 + 1] = (float)s2;
  }
}
__global__ void gn_bwd_apply_kernel(const float* __restrict__ da, const float* __restrict__ x, const float* __restrict__ dres,
                                    const float* __restrict__ stats, const float* __restrict__ S, const float* __restrict__ gamma,
                                    const float* __restrict__ beta, float* __restrict__ dx, size_t total, int HW, int C, int G, int act, float inv_count) {
  const int cpg = C / G;
  const float inv_m = inv_count > 0.f ? inv_count : 1.0f / ((float)cpg * (float)HW);
  GSTRIDE(i, total) {
    const int c = (int)(i % C);
    const size_t n = i / ((size_t)HW * C);
    const size_t gi = (n * G + c / cpg) * 2;
    const float mu = stats[gi], rs = stats[gi + 1];
    const float xh = (x[i] - mu) * rs;
    const float dy = da[i] * act_grad(xh * gamma[c] + beta[c], act);
    const float r = rs * (dy * gamma[c] - S[gi] * inv_m - xh * S[gi + 1] * inv_m);
    dx[i] = dres ? dres[i] + r : r;
  }
}
__global__ void gn_bwd_param_kernel(const float* __restrict__ AB, float* __restrict__ dgamma, float* __restrict__ dbeta, int N, int C, int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double a = 0.0, b = 0.0;
  for (int n = 0; n < N; n++) { a += AB[((size_t)n * C + c) * 2]; b += AB[((size_t)n * C + c) * 2 + 1]; }
  dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)a;
  dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)b;
}

// ---- elementwise family -----------------------------------------------------------------------------------------------------------------
// op 0: out = act(a + b)                 (b may be null; act: 0 none, 1 SiLU, 2 ReLU, 4 LeakyReLU(param); 3: out = b > 0 ? a : 0, the ReLU gate)
// op 1: out = SiLU(a)                    op 2: out = b * SiLU'(a)                       (a = x, b = dy)
// op 3: out = a * (b > 0 ? 1 : param)    (a = dy, b = the activation's OUTPUT: ReLU / LeakyReLU backward)
// op 4: out = GELU(a) (erf form)         op 5: out = b * GELU'(a)
// op 6: out = a + b * g[i % cols]        (LayerScale + residual)                        op 7: out = a * param
__global__ void eltwise_kernel(int op, const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ g, float* __restrict__ out,
                               size_t total, int cols, int act, float param) {
  GSTRIDE(i, total) {
    const float x = a[i];
    float r;
    switch (op) {
      case 0: {
        if (act == 3) { r = b[i] > 0.f ? x : 0.f; break; }
        const float t = b ? x + b[i] : x;
        r = act == 1 ? t * sigmoid_exact(t) : (act == 2 ? fmaxf(t, 0.f) : (act == 4 ? (t > 0.f ? t : param * t) : t));
        break;
      }
      case 1: r = x * sigmoid_exact(x); break;
      case 2: { const float sg = sigmoid_exact(x); r = b[i] * sg * (1.f + x * (1.f - sg)); break; }
      case 3: r = b[i] > 0.f ? x : param * x; break;
      case 4: r = 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); break;
      case 5: r = b[i] * (0.5f * (1.f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * expf(-0.5f * x * x)); break;
      case 6: r = x + b[i] * g[i % cols]; break;
      default: r = x * param; break;
    }
    out[i] = r;
  }
}

// ---- softmax over rows, f32 in / f32 out ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ s, float* __restrict__ p, int rows, int cols, float scale) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* sr = s + (size_t)row * cols;
  float m = -INFINITY;
  for (int c = lane; c < cols; c += 64) m = fmaxf(m, sr[c] * scale);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  double sum = 0.0;
  for (int c = lane; c < cols; c += 64) sum += (double)expf(sr[c] * scale - m);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  const float inv = (float)(1.0 / sum);
  float* pr = p + (size_t)row * cols;
  for (int c = lane; c < cols; c += 64) pr[c] = expf(sr[c] * scale - m) * inv;
}
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ dp, const float* __restrict__ p, float* __restrict__ ds, int rows,
                                                          int cols, float scale) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* dr = dp + (size_t)row * cols;
  const float* pr = p + (size_t)row * cols;
  double dot = 0.0;
  for (int c = lane; c < cols; c += 64) dot += (double)dr[c] * pr[c];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
  float* o = ds + (size_t)row * cols;
  for (int c = lane; c < cols; c += 64) o[c] = scale * pr[c] * (dr[c] - (float)dot);
}

// ---- 2x2 pools on NHWC f32 ---------------------------------------------------------------------------------------------------------------
// op 0: sum pool (backward of nearest x2); op 1: max pool; op 2: max-pool backward fused with the ReLU mask and an extra gradient
// (same contract as misc.hip::maxpool2x2_relu_bwd_kernel: first maximum in scan order takes the pooled gradient)
__global__ void pool2x2_kernel(int op, const float* __restrict__ a, const float* __restrict__ x, const float* __restrict__ extra, float* __restrict__ out,
                               int N, int H, int W, int C) {
  const size_t total = (size_t)N * H * W * C;      // H, W = pooled size
  GSTRIDE(i, total) {
    const int c = (int)(i % C);
    size_t r = i / C;
    const int xo = (int)(r % W); r /= W;
    const int yo = (int)(r % H);
    const size_t n = r / H;
    const size_t b0 = ((n * 2 * H + 2 * yo) * 2 * W + 2 * xo) * C + c;
    const size_t offs[4] = {b0, b0 + C, b0 + (size_t)2 * W * C, b0 + (size_t)2 * W * C + C};
    if (op == 0) {
      out[i] = (a[offs[0]] + a[offs[1]]) + (a[offs[2]] + a[offs[3]]);
    } else if (op == 1) {
      float m = a[offs[0]];
      for (int k = 1; k < 4; k++) m = a[offs[k]] > m ? a[offs[k]] : m;
      out[i] = m;
    } else {
      float m = x[offs[0]];
      int arg = 0;
      for (int k = 1; k < 4; k++) if (x[offs[k]] > m) { m = x[offs[k]]; arg = k; }
      const float g = a ? a[i] : 0.f;
      for (int k = 0; k < 4; k++) {
        const float t = (arg == k ? g : 0.f) + (extra ? extra[offs[k]] : 0.f);
        out[offs[k]] = x[offs[k]] > 0.f ? t : 0.f;
      }
    }
  }
}

// ---- layout ----------------------------------------------------------------------------------------------------------------------------
__global__ void nchw_to_nhwc_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int C, int HW, int Cpad) {
  const size_t total = (size_t)N * HW * Cpad;
  GSTRIDE(i, total) {
    const int c = (int)(i % Cpad);
    const size_t px = i / Cpad;
    const size_t n = px / HW, p = px % HW;
    dst[i] = c < C ? src[(n * C + c) * HW + p] : 0.f;
  }
}

// ---- LPIPS feature difference, f32 features ------------------------------------------------------------------------------------------------
// one wave per pixel: v = sum_c w_c (f0_c/(|f0|+eps) - f1_c/(|f1|+eps))^2 ; df1 = gscale * dv/df1
__global__ __launch_bounds__(256) void lpips_diff_kernel(const float* __restrict__ f0, const float* __restrict__ f1, const float* __restrict__ w,
                                                         float* __restrict__ df1, double* __restrict__ part, size_t pixels, int C, float gscale, float eps) {
  __shared__ double sh[256];
  const int lane = threadIdx.x & 63;
  double acc = 0.0;
  for (size_t p = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); p < pixels; p += (size_t)gridDim.x * 4) {
    const float* a = f0 + p * C;
    const float* b = f1 + p * C;
    float sa = 0.f, sb = 0.f;
    for (int c = lane; c < C; c += 64) { sa += a[c] * a[c]; sb += b[c] * b[c]; }
    sa = wave_sum(sa); sb = wave_sum(sb);
    const float ra = sqrtf(sa), rb = sqrtf(sb);
    const float ia = 1.f / (ra + eps), ib = 1.f / (rb + eps);
    float v = 0.f, gdot = 0.f;
    for (int c = lane; c < C; c += 64) {
#pragma clang fp contract(off)
      const float pa = a[c] * ia, pb = b[c] * ib;
      const float d = pa - pb;
      v += w[c] * d * d;
      gdot += -2.f * w[c] * d * b[c];
    }
    v = wave_sum(v);
    if (df1) {
      gdot = wave_sum(gdot);
      const float k2 = rb > 0.f ? gdot * ib * ib / rb : 0.f;
      for (int c = lane; c < C; c += 64) {
#pragma clang fp contract(off)
        const float pa = a[c] * ia, pb = b[c] * ib;
        const float g = -2.f * w[c] * (pa - pb);
        df1[p * C + c] = gscale * (g * ib - b[c] * k2);
      }
    }
    if (lane == 0) acc += (double)v;
  }
  acc = block_sum_d(acc, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = acc;
}
__global__ void sum_parts_kernel(const double* __restrict__ part, float* __restrict__ out, int n, double scale, int accumulate) {
  double a = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) a += part[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
  if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.f) + (float)(a * scale);
}

// ---- LayerNorm over the last dim, f32 in / f32 out, f64 statistics (timm ViT blocks) ---------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float* __restrict__ y, int rows, int cols, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (size_t)row * cols;
  double s = 0.0, ss = 0.0;
  for (int c = lane; c < cols; c += 64) { const double v = xr[c]; s += v; ss += v * v; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); ss += __shfl_xor(ss, o, 64); }
  const double mean = s / cols;
  double var = ss / cols - mean * mean;
  if (var < 0) var = 0;
  const float mu = (float)mean, rs = (float)(1.0 / sqrt(var + (double)eps));
  float* yr = y + (size_t)row * cols;
  for (int c = lane; c < cols; c += 64) yr[c] = (xr[c] - mu) * rs * gamma[c] + beta[c];
}

}  // namespace dmvae_parity
using namespace dmvae_parity;

extern "C" int dmvae_split3_bf16(const void* x, void* out, size_t rows, int cols, size_t rows_per_batch, size_t batch_stride, size_t part_stride,
                                 size_t row_stride, int pattern, hipStream_t stream) {
  DMVAE_CHECK_ARG(x && out, "split3_bf16: null pointer");
  DMVAE_CHECK_ARG(rows > 0 && cols > 0 && rows_per_batch > 0 && (pattern == 0 || pattern == 1), "split3_bf16: bad shape / pattern");
  const size_t total = rows * (size_t)cols;
  hipLaunchKernelGGL(split3_kernel, dim3(grid_for(total)), dim3(256), 0, stream, (const float*)x, (bf16*)out, total, cols, rows_per_batch, batch_stride,
                     part_stride, row_stride, pattern);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

static int gn_ok(int n, int hw, int c, int groups) { return n > 0 && hw > 0 && c > 0 && groups > 0 && c % groups == 0 && 256 % (c / groups) == 0; }

extern "C" int dmvae_groupnorm_stats_f32(const void* x, void* stats, int n, int hw, int c, int groups, float eps, hipStream_t stream) {
  DMVAE_CHECK_ARG(x && stats, "groupnorm_stats_f32: null pointer");
  DMVAE_CHECK_ARG(gn_ok(n, hw, c, groups), "groupnorm_stats_f32: unsupported shape n=%d hw=%d c=%d groups=%d", n, hw, c, groups);
  hipLaunchKernelGGL(gn_stats_kernel, dim3(groups, n), dim3(256), 0, stream, (const float*)x, (float*)stats, hw, c, groups, eps);
  DMVAE_CHECK_LAUNCH();
  return 0;
}
extern "C" int dmvae_groupnorm_apply_f32(const void* x, const void* stats, const void* gamma, const void* beta, void* y, int n, int hw, int c,
                                         int groups, int act, hipStream_t stream) {
  DMVAE_CHECK_ARG(x && stats && gamma && beta && y, "groupnorm_apply_f32: null pointer");
  DMVAE_CHECK_ARG(gn_ok(n, hw, c, groups) && act >= 0 && act <= 2, "groupnorm_apply_f32: unsupported shape / act");
  const size_t total = (size_t)n * hw * c;
  hipLaunchKernelGGL(gn_apply_kernel, dim3(grid_for(total)), dim3(256), 0, stream, (const float*)x, (const float*)stats, (const float*)gamma,
                     (const float*)beta, (float*)y, total, hw, c, groups, act);
  DMVAE_CHECK_LAUNCH();
  return 0;
}
extern "C" size_t dmvae_groupnorm_f32_workspace(int n, int c, int groups) { return ((size_t)n * c * 2 + (size_t)n * groups * 2) * sizeof(float); }
extern "C" int dmvae_groupnorm_bwd_f32(const void* da, const void* x, const void* dres, const void* stats, const void* gamma, const void* beta, void* dx,
                                       void* dgamma, void* dbeta, void* workspace, size_t workspace_bytes, int n, int hw, int c, int groups, int act,
                                       int accumulate, float inv_count, hipStream_t stream) {
  DMVAE_CHECK_ARG(da && x && stats && gamma && beta && dx && workspace, "groupnorm_bwd_f32: null pointer");
  DMVAE_CHECK_ARG(gn_ok(n, hw, c, groups) && act >= 0 && act <= 2, "groupnorm_bwd_f32: unsupported shape / act");
  DMVAE_CHECK_ARG(workspace_bytes >= dmvae_groupnorm_f32_workspace(n, c, groups), "groupnorm_bwd_f32: workspace too small");
  float* AB = (float*)workspace;
  float* S = AB + (size_t)n * c * 2;
  hipLaunchKernelGGL(gn_bwd_reduce_kernel, dim3(groups, n), dim3(256), 0, stream, (const float*)da, (const float*)x, (const float*)stats,
                     (const float*)gamma, (const float*)beta, AB, S, hw, c, groups, act);
  DMVAE_CHECK_LAUNCH();
  const size_t total = (size_t)n * hw * c;
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(grid_for(total)), dim3(256), 0, stream, (const float*)da, (const float*)x, (const float*)dres,
                     (const float*)stats, S, (const float*)gamma, (const float*)beta, (float*)dx, total, hw, c, groups, act, inv_count);
  DMVAE_CHECK_LAUNCH();
  if (dgamma && dbeta) {
    hipLaunchKernelGGL(gn_bwd_param_kernel, dim3((c + 255) / 256), dim3(256), 0, stream, AB, (float*)dgamma, (float*)dbeta, n, c, accumulate);
    DMVAE_CHECK_LAUNCH();
  }
  return 0;
}

extern "C" int dmvae_eltwise_f32(int op, const void* a, const void* b, const void* g, void* out, size_t n, int cols, int act, float param,
                                 hipStream_t stream) {
  DMVAE_CHECK_ARG(a && out && n > 0, "eltwise_f32: null pointer / empty");
  DMVAE_CHECK_ARG(op >= 0 && op <= 7, "eltwise_f32: unknown op %d", op);
  DMVAE_CHECK_ARG(!(op == 2 || op == 3 || op == 5 || op == 6 || (op == 0 && act == 3)) || b, "eltwise_f32: op %d needs a second operand", op);
  DMVAE_CHECK_ARG(op != 6 || (g && cols > 0), "eltwise_f32: op 6 needs the per-column scale");
  hipLaunchKernelGGL(eltwise_kernel, dim3(grid_for(n)), dim3(256), 0, stream, op, (const float*)a, (const float*)b, (const float*)g, (float*)out, n,
                     cols > 0 ? cols : 1, act, param);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

extern "C" int dmvae_softmax_rows_fwd_f32(const void* s, void* p, int rows, int cols, float scale, hipStream_t stream) {
  DMVAE_CHECK_ARG(s && p && rows > 0 && cols > 0, "softmax_rows_fwd_f32: bad arguments");
  hipLaunchKernelGGL(dmvae_parity::softmax_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, (const float*)s, (float*)p, rows, cols, scale);
  DMVAE_CHECK_LAUNCH();
  return 0;
}
extern "C" int dmvae_softmax_rows_bwd_f32(const void* dp, const void* p, void* ds, int rows, int cols, float scale, hipStream_t stream) {
  DMVAE_CHECK_ARG(dp && p && ds && rows > 0 && cols > 0, "softmax_rows_bwd_f32: bad arguments");
  hipLaunchKernelGGL(dmvae_parity::softmax_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, (const float*)dp, (const float*)p, (float*)ds, rows, cols,
                     scale);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

/* n, h, w = POOLED size.  op 0: out[n,h,w,c] = sum of a's 2x2 window; op 1: max; op 2: out[n,2h,2w,c] = ReLU-masked max-pool backward of a (pooled
 * gradient, may be NULL) at the argmax of x's window plus `extra` (may be NULL). */
extern "C" int dmvae_pool2x2_f32(int op, const void* a, const void* x, const void* extra, void* out, int n, int h, int w, int c, hipStream_t stream) {
  DMVAE_CHECK_ARG(out && op >= 0 && op <= 2 && n > 0 && h > 0 && w > 0 && c > 0, "pool2x2_f32: bad arguments");
  DMVAE_CHECK_ARG(op == 2 ? x != nullptr : a != nullptr, "pool2x2_f32: null operand");
  hipLaunchKernelGGL(pool2x2_kernel, dim3(grid_for((size_t)n * h * w * c)), dim3(256), 0, stream, op, (const float*)a, (const float*)x, (const float*)extra,
                     (float*)out, n, h, w, c);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

extern "C" int dmvae_nchw_f32_to_nhwc_f32(const void* src, void* dst, int n, int c, int hw, int c_pad, hipStream_t stream) {
  DMVAE_CHECK_ARG(src && dst && n > 0 && c > 0 && hw > 0 && c_pad >= c, "nchw_f32_to_nhwc_f32: bad arguments");
  hipLaunchKernelGGL(nchw_to_nhwc_f32_kernel, dim3(grid_for((size_t)n * hw * c_pad)), dim3(256), 0, stream, (const float*)src, (float*)dst, n, c, hw, c_pad);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

extern "C" int dmvae_lpips_diff_f32(const void* f0, const void* f1, const void* lin_w, void* df1, void* out, void* workspace, size_t workspace_bytes, int n,
                                    int hw, int c, float gscale, int accumulate, hipStream_t stream) {
  DMVAE_CHECK_ARG(f0 && f1 && lin_w && out && workspace, "lpips_diff_f32: null pointer");
  DMVAE_CHECK_ARG(n > 0 && hw > 0 && c > 0, "lpips_diff_f32: bad shape");
  const size_t pixels = (size_t)n * hw;
  int blocks = (int)((pixels + 3) / 4);
  if (blocks > 2048) blocks = 2048;
  DMVAE_CHECK_ARG(workspace_bytes >= (size_t)blocks * sizeof(double), "lpips_diff_f32: workspace too small");
  hipLaunchKernelGGL(dmvae_parity::lpips_diff_kernel, dim3(blocks), dim3(256), 0, stream, (const float*)f0, (const float*)f1, (const float*)lin_w,
                     (float*)df1, (double*)workspace, pixels, c, gscale, 1e-10f);
  DMVAE_CHECK_LAUNCH();
  hipLaunchKernelGGL(sum_parts_kernel, dim3(1), dim3(64), 0, stream, (const double*)workspace, (float*)out, blocks, 1.0 / ((double)hw * n), accumulate);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

extern "C" int dmvae_layernorm_f32(const void* x, const void* gamma, const void* beta, void* y, int rows, int cols, float eps, hipStream_t stream) {
  DMVAE_CHECK_ARG(x && gamma && beta && y && rows > 0 && cols > 0, "layernorm_f32: bad arguments");
  hipLaunchKernelGGL(dmvae_parity::layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, (const float*)x, (const float*)gamma, (const float*)beta,
                     (float*)y, rows, cols, eps);
  DMVAE_CHECK_LAUNCH();
  return 0;
}
