// Elementwise kernels of the LightningDiT inference path (diffusion/lightningdit/lightningdit.py:175-273 with RMSNorm, QK-norm, RoPE, SwiGLU and
// shift / scale / gate adaLN; evaluated four times per VAE turn inside train_dmd.py's DMD loss, :211-217).  Residual stream f32 (the f32 position
// table promotes it), Linear operands / results bf16 (autocast); every kernel rounds to bf16 exactly where the reference's autocast graph does.
//
//   rmsnorm_modulate : a = bf16( x * rsqrt(mean x^2 + eps) * w * (1 + scale[b]) + shift[b] )                 (norm1 / norm2 / norm_final + modulate)
//   qknorm_rope      : per (token, head): q, k -> bf16(RMSNorm) * w -> 2-D rotary embedding -> bf16, written head-major [B*H][N][Dp] with the
//                      head dim zero-padded to a multiple of 32 (the batched-GEMM kernel's K step); v copied head-major
//   swiglu           : h = bf16( bf16(silu(x1)) * x2 )  for [x1, x2] = w12(a)
//   gated_residual   : x += bf16( gate[b] * y )
//
// All HBM-bound single passes with 8- or 16-byte accesses.  GEMMs go through the library, attention through the batched GEMM + softmax kernels.
#include "common.h"
#include <cstdlib>
#include <type_traits>
#include "dmvae_hip.h"

namespace dmvae_dit {

constexpr int MAX_SWEEPS = 8;  // C <= 2048

// one wave per row; mod: [B][stride] bf16 (the adaLN Linear's output), shift_off < 0: no shift
// RES: first x[row] += bf16(gate[b] * r[row]) (the previous sub-layer's gated residual, written back), then the norm of the updated row.
// SW = ceil(C / 256) sweeps per row (register arrays sized by the width: 5 at LightningDiT-XL's 1152)
template <bool RES, int SW>
__global__ __launch_bounds__(256) void rmsnorm_modulate_kernel(const float* x, const float* __restrict__ w, const bf16* __restrict__ mod,
                                                               bf16* __restrict__ y, int rows, int rows_per_sample, int C, int stride, int shift_off,
                                                               int scale_off, float eps, const bf16* __restrict__ r, const bf16* __restrict__ gmod,
                                                               int gstride, int gate_off, float* xo) {   // RES: updated rows go to xo (== x: in place); y == null: no norm
#pragma clang fp contract(off)   // both instantiations evaluate the expressions exactly as written: fused and unfused routes agree to the bit
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (size_t)row * C;
  f32x4 v[SW];
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < SW; k++) {
    const int c = k * 256 + lane * 4;
    if (c < C) {
      v[k] = *reinterpret_cast<const f32x4*>(xr + c);
      if constexpr (RES) {
        const bf16x4 rv = *reinterpret_cast<const bf16x4*>(r + (size_t)row * C + c);
        const bf16x4 g = *reinterpret_cast<const bf16x4*>(gmod + (size_t)(row / rows_per_sample) * gstride + gate_off + c);
#pragma unroll
        for (int e = 0; e < 4; e++) v[k][e] += (float)(bf16)((float)g[e] * (float)rv[e]);
        *reinterpret_cast<f32x4*>(xo + (size_t)row * C + c) = v[k];
      }
      ss += (v[k][0] * v[k][0] + v[k][1] * v[k][1]) + (v[k][2] * v[k][2] + v[k][3] * v[k][3]);
    }
  }
  if (RES && !y) return;                       // gated residual only (wave-uniform)
  // the norm's own operands (weight, scale, shift: L2-resident vectors) are requested BEFORE the row reduction: their round trip overlaps the butterfly instead of
  // following it (one memory latency less on a kernel that is a single round of waves)
  const bf16* mrow = mod + (size_t)(row / rows_per_sample) * stride;
  f32x4 gv[SW];
  bf16x4 scv[SW], shv[SW];
#pragma unroll
  for (int k = 0; k < SW; k++) {
    const int c = k * 256 + lane * 4;
    if (c < C) {
      gv[k] = *reinterpret_cast<const f32x4*>(w + c);
      scv[k] = *reinterpret_cast<const bf16x4*>(mrow + scale_off + c);
      shv[k] = bf16x4{(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
      if (shift_off >= 0) shv[k] = *reinterpret_cast<const bf16x4*>(mrow + shift_off + c);
    }
  }
  const float rs = rsqrtf(wave_sum(ss) / (float)C + eps);
  bf16* yr = y + (size_t)row * C;
#pragma unroll
  for (int k = 0; k < SW; k++) {
    const int c = k * 256 + lane * 4;
    if (c < C) {
      const f32x4 g = gv[k];
      const bf16x4 sc = scv[k], sh = shv[k];
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; e++)  // `1 + scale` is a bf16 tensor in the reference's autocast graph (bf16 scale): rounded before it multiplies
        o[e] = (bf16)(v[k][e] * rs * g[e] * (float)(bf16)(1.f + (float)sc[e]) + (float)sh[e]);
      *reinterpret_cast<bf16x4*>(yr + c) = o;
    }
  }
}

// The same kernel with EIGHT channels per lane (C % 8 == 0): 32-B f32 and 16-B bf16 accesses per lane -- half the memory instructions, and the bf16 result leaves
// in 16-B stores (1 KiB per wave-instruction; the 4-channel form's 8-B stores are store-issue-bound on the large calls: 2.95 TB/s at batch 64).
// SW8 = ceil(C / 512).  Per-row arithmetic as above; the sum of squares is taken in this kernel's own lane order, so the two forms differ in the last f32 bit of rs.
template <bool RES, int SW8>
__global__ __launch_bounds__(256) void rmsnorm_modulate8_kernel(const float* x, const float* __restrict__ w, const bf16* __restrict__ mod,
                                                                bf16* __restrict__ y, int rows, int rows_per_sample, int C, int stride, int shift_off,
                                                                int scale_off, float eps, const bf16* __restrict__ r, const bf16* __restrict__ gmod,
                                                                int gstride, int gate_off, float* xo) {
#pragma clang fp contract(off)
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (size_t)row * C;
  f32x4 v0[SW8], v1[SW8];
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < SW8; k++) {
    const int c = k * 512 + lane * 8;
    if (c < C) {
      v0[k] = *reinterpret_cast<const f32x4*>(xr + c);
      v1[k] = *reinterpret_cast<const f32x4*>(xr + c + 4);
      if constexpr (RES) {
        const bf16x8 rv = *reinterpret_cast<const bf16x8*>(r + (size_t)row * C + c);
        const bf16x8 g = *reinterpret_cast<const bf16x8*>(gmod + (size_t)(row / rows_per_sample) * gstride + gate_off + c);
#pragma unroll
        for (int e = 0; e < 4; e++) {
          v0[k][e] += (float)(bf16)((float)g[e] * (float)rv[e]);
          v1[k][e] += (float)(bf16)((float)g[4 + e] * (float)rv[4 + e]);
        }
        *reinterpret_cast<f32x4*>(xo + (size_t)row * C + c) = v0[k];
        *reinterpret_cast<f32x4*>(xo + (size_t)row * C + c + 4) = v1[k];
      }
      ss += ((v0[k][0] * v0[k][0] + v0[k][1] * v0[k][1]) + (v0[k][2] * v0[k][2] + v0[k][3] * v0[k][3])) +
            ((v1[k][0] * v1[k][0] + v1[k][1] * v1[k][1]) + (v1[k][2] * v1[k][2] + v1[k][3] * v1[k][3]));
    }
  }
  if (RES && !y) return;
  const bf16* mrow = mod + (size_t)(row / rows_per_sample) * stride;
  f32x4 g0[SW8], g1[SW8];
  bf16x8 scv[SW8], shv[SW8];
#pragma unroll
  for (int k = 0; k < SW8; k++) {
    const int c = k * 512 + lane * 8;
    if (c < C) {
      g0[k] = *reinterpret_cast<const f32x4*>(w + c);
      g1[k] = *reinterpret_cast<const f32x4*>(w + c + 4);
      scv[k] = *reinterpret_cast<const bf16x8*>(mrow + scale_off + c);
#pragma unroll
      for (int e = 0; e < 8; e++) shv[k][e] = (bf16)0.f;
      if (shift_off >= 0) shv[k] = *reinterpret_cast<const bf16x8*>(mrow + shift_off + c);
    }
  }
  const float rs = rsqrtf(wave_sum(ss) / (float)C + eps);
  bf16* yr = y + (size_t)row * C;
#pragma unroll
  for (int k = 0; k < SW8; k++) {
    const int c = k * 512 + lane * 8;
    if (c < C) {
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        o[e] = (bf16)(v0[k][e] * rs * g0[k][e] * (float)(bf16)(1.f + (float)scv[k][e]) + (float)shv[k][e]);
        o[4 + e] = (bf16)(v1[k][e] * rs * g1[k][e] * (float)(bf16)(1.f + (float)scv[k][4 + e]) + (float)shv[k][4 + e]);
      }
      *reinterpret_cast<bf16x8*>(yr + c) = o;
    }
  }
}

// one wave per (token, head); lane j < D/2 owns the feature pair (2j, 2j+1)
__global__ __launch_bounds__(256) void qknorm_rope_kernel(const bf16* __restrict__ qkv, const float* __restrict__ qw, const float* __restrict__ kw,
                                                          const float* __restrict__ cosb, const float* __restrict__ sinb, bf16* __restrict__ qo,
                                                          bf16* __restrict__ ko, bf16* __restrict__ vo, int tokens, int N, int H, int D, int Dp,
                                                          float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= tokens * H) return;
  const int tok = row / H, h = row - tok * H;
  const int b = tok / N, n = tok - b * N;
  const bool live = 2 * lane < D, pad = 2 * lane >= D && 2 * lane < Dp;
  const bf16* base = qkv + (size_t)tok * 3 * H * D;
  float q0 = 0.f, q1 = 0.f, k0 = 0.f, k1 = 0.f, c0 = 0.f, s0 = 0.f, c1 = 0.f, s1 = 0.f, wq0 = 0.f, wq1 = 0.f, wk0 = 0.f, wk1 = 0.f;
  bf16x2 vv = {(bf16)0.f, (bf16)0.f};
  if (live) {
    const bf16x2 a = *reinterpret_cast<const bf16x2*>(base + (size_t)h * D + 2 * lane);
    const bf16x2 c = *reinterpret_cast<const bf16x2*>(base + (size_t)(H + h) * D + 2 * lane);
    vv = *reinterpret_cast<const bf16x2*>(base + (size_t)(2 * H + h) * D + 2 * lane);
    q0 = (float)a[0]; q1 = (float)a[1]; k0 = (float)c[0]; k1 = (float)c[1];
    c0 = cosb[(size_t)n * D + 2 * lane]; c1 = cosb[(size_t)n * D + 2 * lane + 1];
    s0 = sinb[(size_t)n * D + 2 * lane]; s1 = sinb[(size_t)n * D + 2 * lane + 1];
    wq0 = qw[2 * lane]; wq1 = qw[2 * lane + 1]; wk0 = kw[2 * lane]; wk1 = kw[2 * lane + 1];
  }
  const float rq = rsqrtf(wave_sum(q0 * q0 + q1 * q1) / (float)D + eps);
  const float rk = rsqrtf(wave_sum(k0 * k0 + k1 * k1) / (float)D + eps);
  // RMSNorm casts back to the input dtype (bf16) before the f32 weight multiplies (rms_norm.py:75-76)
  const float nq0 = (float)(bf16)(q0 * rq) * wq0, nq1 = (float)(bf16)(q1 * rq) * wq1;
  const float nk0 = (float)(bf16)(k0 * rk) * wk0, nk1 = (float)(bf16)(k1 * rk) * wk1;
  const size_t o = ((size_t)(b * H + h) * N + n);
  if (live) {
    const bf16x2 qq = {(bf16)(nq0 * c0 - nq1 * s0), (bf16)(nq1 * c1 + nq0 * s1)};
    const bf16x2 kk = {(bf16)(nk0 * c0 - nk1 * s0), (bf16)(nk1 * c1 + nk0 * s1)};
    *reinterpret_cast<bf16x2*>(qo + o * Dp + 2 * lane) = qq;
    *reinterpret_cast<bf16x2*>(ko + o * Dp + 2 * lane) = kk;
    *reinterpret_cast<bf16x2*>(vo + o * D + 2 * lane) = vv;
  } else if (pad) {
    const bf16x2 z = {(bf16)0.f, (bf16)0.f};
    *reinterpret_cast<bf16x2*>(qo + o * Dp + 2 * lane) = z;
    *reinterpret_cast<bf16x2*>(ko + o * Dp + 2 * lane) = z;
  }
}

__global__ void swiglu_kernel(const bf16* __restrict__ x12, bf16* __restrict__ out, size_t rows, int hid8) {
  const size_t total = rows * hid8;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / hid8;
    const int c = (int)(i - r * hid8);
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(x12 + (r * 2 * hid8 + c) * 8);
    const bf16x8 b = *reinterpret_cast<const bf16x8*>(x12 + (r * 2 * hid8 + hid8 + c) * 8);
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const float x = (float)a[e];
      o[e] = (bf16)((float)(bf16)(x * sigmoidf_(x)) * (float)b[e]);
    }
    reinterpret_cast<bf16x8*>(out)[i] = o;
  }
}

__global__ void gated_residual_kernel(float* __restrict__ x, const bf16* __restrict__ y, const bf16* __restrict__ mod, size_t rows, int c8,
                                      int rows_per_sample, int stride, int gate_off) {
  const size_t total = rows * c8;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / c8;
    const int c = (int)(i - r * c8) * 8;
    const bf16x8 v = reinterpret_cast<const bf16x8*>(y)[i];
    const bf16x8 g = *reinterpret_cast<const bf16x8*>(mod + (r / rows_per_sample) * stride + gate_off + c);
    f32x4 a = reinterpret_cast<const f32x4*>(x)[2 * i], b = reinterpret_cast<const f32x4*>(x)[2 * i + 1];
#pragma unroll
    for (int e = 0; e < 4; e++) {
      a[e] += (float)(bf16)((float)g[e] * (float)v[e]);
      b[e] += (float)(bf16)((float)g[4 + e] * (float)v[4 + e]);
    }
    reinterpret_cast<f32x4*>(x)[2 * i] = a;
    reinterpret_cast<f32x4*>(x)[2 * i + 1] = b;
  }
}

// ---- backward side (the student's training turn, train_dmd.py:565-575) --------------------------------------------------------------------
// Gradients are bf16 where the forward tensor is bf16 and f32 on the residual stream; per-sample reductions (d shift / d scale / d gate of the
// adaLN chunks) and per-channel reductions (norm weights) are two-stage with a fixed order.

// gated residual x_out = x + bf16(gate[b] * y):  dy = bf16(gate[b] * dx_out),  dgate[b][c] = sum_n dx_out * y.   grid (C/256, B); block = 32 channel
// octets x 8 token lanes
__global__ __launch_bounds__(256) void gated_residual_bwd_kernel(const float* __restrict__ dx, const bf16* __restrict__ y, const bf16* __restrict__ mod,
                                                                 bf16* __restrict__ dy, float* __restrict__ dmod, int N, int C, int stride, int gate_off) {
  __shared__ float red[8][256];
  const int b = blockIdx.y, oc = threadIdx.x & 31, tl = threadIdx.x >> 5;
  const int c = blockIdx.x * 256 + oc * 8;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (c < C) {
    const bf16x8 g = *reinterpret_cast<const bf16x8*>(mod + (size_t)b * stride + gate_off + c);
    for (int n = tl; n < N; n += 8) {
      const size_t off = ((size_t)b * N + n) * C + c;
      const f32x4 d0 = *reinterpret_cast<const f32x4*>(dx + off), d1 = *reinterpret_cast<const f32x4*>(dx + off + 4);
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(y + off);
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        o[e] = (bf16)((float)g[e] * d0[e]); o[4 + e] = (bf16)((float)g[4 + e] * d1[e]);
        acc[e] += d0[e] * (float)v[e]; acc[4 + e] += d1[e] * (float)v[4 + e];
      }
      *reinterpret_cast<bf16x8*>(dy + off) = o;
    }
  }
#pragma unroll
  for (int e = 0; e < 8; e++) red[tl][oc * 8 + e] = acc[e];
  __syncthreads();
  const int cc = blockIdx.x * 256 + threadIdx.x;
  if (cc < C) {
    float a = 0.f;
#pragma unroll
    for (int q = 0; q < 8; q++) a += red[q][threadIdx.x];
    dmod[(size_t)b * stride + gate_off + cc] = a;
  }
}

__device__ __forceinline__ float silu_grad_(float x) { const float s = sigmoidf_(x); return s * (1.f + x * (1.f - s)); }

// h = bf16(bf16(silu(x1)) * x2):  dx1 = dh * x2 * silu'(x1),  dx2 = dh * bf16(silu(x1))
__global__ void swiglu_bwd_kernel(const bf16* __restrict__ dh, const bf16* __restrict__ x12, bf16* __restrict__ dx12, size_t rows, int hid8) {
  const size_t total = rows * hid8;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / hid8;
    const int c = (int)(i - r * hid8);
    const size_t o1 = (r * 2 * hid8 + c) * 8, o2 = (r * 2 * hid8 + hid8 + c) * 8;
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(x12 + o1), b = *reinterpret_cast<const bf16x8*>(x12 + o2);
    const bf16x8 d = reinterpret_cast<const bf16x8*>(dh)[i];
    bf16x8 g1, g2;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const float x = (float)a[e], dv = (float)d[e];
      g1[e] = (bf16)(dv * (float)b[e] * silu_grad_(x));
      g2[e] = (bf16)(dv * (float)(bf16)(x * sigmoidf_(x)));
    }
    *reinterpret_cast<bf16x8*>(dx12 + o1) = g1;
    *reinterpret_cast<bf16x8*>(dx12 + o2) = g2;
  }
}

// a = bf16( n * w * m + shift ),  n = x * rs,  m = bf16(1 + scale[b]):
//   dx += rs * (g - n * mean(g * n)),  g = da * w * m;   dshift[b] = sum_n da;   dscale[b] = sum_n da * n * w;   dw = sum_rows da * m * n.
// grid (BPS, B): the block's 4 waves walk the rows of sample b; part: [B][BPS][3][C] (dshift, dscale, dw contributions of the block)
constexpr int RM_MAX_SEQ = 4096;   // row statistics of the two-kernel RMSNorm backward live in the workspace up to this many tokens per sample
constexpr int RM_BPS = 32;   // blocks per sample: 4 waves x 2 rows each at N = 256 tokens
// SW = ceil(C / 256) sweeps per row: seven f32x4 arrays of that length live in registers (224 VGPRs at the 2048-channel maximum, 140 at 1152)
template <int SW>
__global__ __launch_bounds__(256) void rmsnorm_modulate_bwd_kernel(const bf16* __restrict__ da, const float* __restrict__ x, const float* __restrict__ w,
                                                                   const bf16* __restrict__ mod, float* __restrict__ dx_io, float* __restrict__ part,
                                                                   int N, int C, int stride, int scale_off, float eps) {
  extern __shared__ float red[];  // [2 wave pairs][3][C]
  const int b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  f32x4 gw[SW], gm[SW], a0[SW], a1[SW], a2[SW];
#pragma unroll
  for (int k = 0; k < SW; k++) {
    const int c = k * 256 + lane * 4;
    a0[k] = f32x4{0, 0, 0, 0}; a1[k] = f32x4{0, 0, 0, 0}; a2[k] = f32x4{0, 0, 0, 0};
    if (c < C) {
      gw[k] = *reinterpret_cast<const f32x4*>(w + c);
      const bf16x4 sc = *reinterpret_cast<const bf16x4*>(mod + (size_t)b * stride + scale_off + c);
#pragma unroll
      for (int e = 0; e < 4; e++) gm[k][e] = (float)(bf16)(1.f + (float)sc[e]);
    }
  }
  for (int n = blockIdx.x * 4 + wave; n < N; n += gridDim.x * 4) {
    const size_t row = (size_t)b * N + n;
    f32x4 v[SW], g[SW];
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < SW; k++) {
      const int c = k * 256 + lane * 4;
      if (c < C) {
        v[k] = *reinterpret_cast<const f32x4*>(x + row * C + c);
        ss += (v[k][0] * v[k][0] + v[k][1] * v[k][1]) + (v[k][2] * v[k][2] + v[k][3] * v[k][3]);
      }
    }
    const float rs = rsqrtf(wave_sum(ss) / (float)C + eps);
    float s2 = 0.f;
#pragma unroll
    for (int k = 0; k < SW; k++) {
      const int c = k * 256 + lane * 4;
      if (c < C) {
        const bf16x4 d = *reinterpret_cast<const bf16x4*>(da + row * C + c);
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const float nh = v[k][e] * rs, dv = (float)d[e];
          v[k][e] = nh;
          g[k][e] = dv * gw[k][e] * gm[k][e];
          s2 += g[k][e] * nh;
          a0[k][e] += dv; a1[k][e] += dv * nh * gw[k][e]; a2[k][e] += dv * gm[k][e] * nh;
        }
      }
    }
    const float m2 = wave_sum(s2) / (float)C;
#pragma unroll
    for (int k = 0; k < SW; k++) {
      const int c = k * 256 + lane * 4;
      if (c < C) {
        f32x4 o = *reinterpret_cast<const f32x4*>(dx_io + row * C + c);
#pragma unroll
        for (int e = 0; e < 4; e++) o[e] += rs * (g[k][e] - v[k][e] * m2);
        *reinterpret_cast<f32x4*>(dx_io + row * C + c) = o;
      }
    }
  }
  // waves 0/1 and 2/3 combine in registers-through-LDS pairs in a fixed order: one [2][3][C] buffer (27 KB at C = 1152) instead of [4][3][C], so that
  // LDS no longer caps the kernel at two blocks per CU
#pragma unroll
  for (int k = 0; k < SW; k++) {
    const int c = k * 256 + lane * 4;
    if (c < C && (wave & 1)) {
      *reinterpret_cast<f32x4*>(&red[((wave >> 1) * 3 + 0) * C + c]) = a0[k];
      *reinterpret_cast<f32x4*>(&red[((wave >> 1) * 3 + 1) * C + c]) = a1[k];
      *reinterpret_cast<f32x4*>(&red[((wave >> 1) * 3 + 2) * C + c]) = a2[k];
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < SW; k++) {
    const int c = k * 256 + lane * 4;
    if (c < C && !(wave & 1)) {
      f32x4* p0 = reinterpret_cast<f32x4*>(&red[((wave >> 1) * 3 + 0) * C + c]);
      f32x4* p1 = reinterpret_cast<f32x4*>(&red[((wave >> 1) * 3 + 1) * C + c]);
      f32x4* p2 = reinterpret_cast<f32x4*>(&red[((wave >> 1) * 3 + 2) * C + c]);
      *p0 = a0[k] + *p0; *p1 = a1[k] + *p1; *p2 = a2[k] + *p2;
    }
  }
  __syncthreads();
  float* po = part + ((size_t)b * gridDim.x + blockIdx.x) * 3 * C;
  for (int i = threadIdx.x; i < 3 * C; i += 256) po[i] = red[i] + red[3 * C + i];
}
// second stage, grid (C/256, B): dmod[b][shift_off + c] = sum_blk part[b][blk][0][c] (skipped when shift_off < 0), dmod[b][scale_off + c] = ... [1] ...;
// wpart[b][c] = sum_blk part[b][blk][2][c];  third stage: dw[c] (+)= sum_b wpart[b][c]
// Two-kernel form of the same backward (used when the row statistics fit the workspace): (A) one wave per row -> rstd and m2 = mean_c(g * xhat),
// g = da * w * bf16(1 + scale); (B) one thread per 4 columns walking the block's rows with fully coalesced row accesses, 12 accumulators per thread and no
// cross-lane work -- the single-pass kernel above keeps 7 x SW f32x4 arrays per lane and two dependent wave reductions per row, and is latency-bound.
__global__ __launch_bounds__(256) void rmsnorm_rowstat_kernel(const bf16* __restrict__ da, const float* __restrict__ x, const float* __restrict__ w,
                                                              const bf16* __restrict__ mod, float2* __restrict__ rowstat, int rows, int N, int C, int stride,
                                                              int scale_off, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const bf16* mrow = mod + (size_t)(row / N) * stride + scale_off;
  float ss = 0.f, s2 = 0.f;
  for (int c = lane * 4; c < C; c += 256) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)row * C + c);
    const bf16x4 d = *reinterpret_cast<const bf16x4*>(da + (size_t)row * C + c);
    const f32x4 gw = *reinterpret_cast<const f32x4*>(w + c);
    const bf16x4 sc = *reinterpret_cast<const bf16x4*>(mrow + c);
#pragma unroll
    for (int e = 0; e < 4; e++) {
      ss += v[e] * v[e];
      s2 += (float)d[e] * gw[e] * (float)(bf16)(1.f + (float)sc[e]) * v[e];
    }
  }
  const float rs = rsqrtf(wave_sum(ss) / (float)C + eps);
  const float m2 = wave_sum(s2) * rs / (float)C;
  if (lane == 0) rowstat[row] = make_float2(rs, m2);
}

__global__ __launch_bounds__(512) void rmsnorm_modulate_bwd_apply_kernel(const bf16* __restrict__ da, const float* __restrict__ x, const float* __restrict__ w,
                                                                         const bf16* __restrict__ mod, const float2* __restrict__ rowstat,
                                                                         float* __restrict__ dx_io, float* __restrict__ part, int N, int C, int stride,
                                                                         int scale_off) {
  const int b = blockIdx.y, c = threadIdx.x * 4;
  if (c >= C) return;
  const f32x4 gw = *reinterpret_cast<const f32x4*>(w + c);
  const bf16x4 sc = *reinterpret_cast<const bf16x4*>(mod + (size_t)b * stride + scale_off + c);
  f32x4 gm, a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0};
#pragma unroll
  for (int e = 0; e < 4; e++) gm[e] = (float)(bf16)(1.f + (float)sc[e]);
  const int rpb = (N + gridDim.x - 1) / gridDim.x;
  const int n1 = min(N, (int)(blockIdx.x + 1) * rpb);
  for (int n = blockIdx.x * rpb; n < n1; n++) {
    const size_t row = (size_t)b * N + n;
    const float2 st = rowstat[row];
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + row * C + c);
    const bf16x4 d = *reinterpret_cast<const bf16x4*>(da + row * C + c);
    f32x4 o = *reinterpret_cast<const f32x4*>(dx_io + row * C + c);
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const float nh = v[e] * st.x, dv = (float)d[e];
      const float g = dv * gw[e] * gm[e];
      o[e] += st.x * (g - nh * st.y);
      a0[e] += dv; a1[e] += dv * nh * gw[e]; a2[e] += dv * gm[e] * nh;
    }
    *reinterpret_cast<f32x4*>(dx_io + row * C + c) = o;
  }
  float* po = part + ((size_t)b * gridDim.x + blockIdx.x) * 3 * C;
  *reinterpret_cast<f32x4*>(po + c) = a0;
  *reinterpret_cast<f32x4*>(po + C + c) = a1;
  *reinterpret_cast<f32x4*>(po + 2 * C + c) = a2;
}

__global__ void rmsnorm_modulate_bwd_final_kernel(const float* __restrict__ part, float* __restrict__ dmod, float* __restrict__ wpart, int bps, int C,
                                                  int stride, int shift_off, int scale_off) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (c >= C) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  for (int k = 0; k < bps; k++) {
    const float* q = part + ((size_t)b * bps + k) * 3 * C;
    s0 += q[c]; s1 += q[C + c]; s2 += q[2 * C + c];
  }
  if (shift_off >= 0) dmod[(size_t)b * stride + shift_off + c] = s0;
  dmod[(size_t)b * stride + scale_off + c] = s1;
  wpart[(size_t)b * C + c] = s2;
}
__global__ void rmsnorm_modulate_bwd_weight_kernel(const float* __restrict__ wpart, float* __restrict__ dw, int B, int C, int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float a = 0.f;
  for (int b = 0; b < B; b++) a += wpart[(size_t)b * C + c];
  dw[c] = (accumulate ? dw[c] : 0.f) + a;
}

// Backward of qknorm_rope: dq', dk' [B*H][N][Dp] bf16 (w.r.t. the rotated, normalised q / k), dv [B*H][N][D] -> dqkv [B][N][3][H][D] bf16 and the
// per-block partial sums of the two norm weights' gradients, part [gridDim.x][2][D].
__global__ __launch_bounds__(256) void qknorm_rope_bwd_kernel(const bf16* __restrict__ dq, const bf16* __restrict__ dk, const bf16* __restrict__ dv,
                                                              const bf16* __restrict__ qkv, const float* __restrict__ qw, const float* __restrict__ kw,
                                                              const float* __restrict__ cosb, const float* __restrict__ sinb, bf16* __restrict__ dqkv,
                                                              float* __restrict__ part, int tokens, int N, int H, int D, int Dp, float eps) {
  __shared__ float red[4][4][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool live = 2 * lane < D;
  float gq0 = 0.f, gq1 = 0.f, gk0 = 0.f, gk1 = 0.f;     // weight-gradient accumulators of this lane's feature pair
  float wq0 = 0.f, wq1 = 0.f, wk0 = 0.f, wk1 = 0.f;
  if (live) { wq0 = qw[2 * lane]; wq1 = qw[2 * lane + 1]; wk0 = kw[2 * lane]; wk1 = kw[2 * lane + 1]; }
  for (int row = blockIdx.x * 4 + wave; row < tokens * H; row += gridDim.x * 4) {
    const int tok = row / H, h = row - tok * H;
    const int b = tok / N, n = tok - b * N;
    float c0 = 0.f, s0 = 0.f, c1 = 0.f, s1 = 0.f;
    if (live) {
      c0 = cosb[(size_t)n * D + 2 * lane]; c1 = cosb[(size_t)n * D + 2 * lane + 1];
      s0 = sinb[(size_t)n * D + 2 * lane]; s1 = sinb[(size_t)n * D + 2 * lane + 1];
    }
    const bf16* base = qkv + (size_t)tok * 3 * H * D;
    bf16* dbase = dqkv + (size_t)tok * 3 * H * D;
    {
      const size_t o = ((size_t)(b * H + h) * N + n);
      float q0 = 0.f, q1 = 0.f, k0 = 0.f, k1 = 0.f, dq0 = 0.f, dq1 = 0.f, dk0 = 0.f, dk1 = 0.f;
      bf16x2 dvv = {(bf16)0.f, (bf16)0.f};
      if (live) {
        const bf16x2 a = *reinterpret_cast<const bf16x2*>(base + (size_t)h * D + 2 * lane);
        const bf16x2 c = *reinterpret_cast<const bf16x2*>(base + (size_t)(H + h) * D + 2 * lane);
        q0 = (float)a[0]; q1 = (float)a[1]; k0 = (float)c[0]; k1 = (float)c[1];
        const bf16x2 ga = *reinterpret_cast<const bf16x2*>(dq + o * Dp + 2 * lane);
        const bf16x2 gc = *reinterpret_cast<const bf16x2*>(dk + o * Dp + 2 * lane);
        dvv = *reinterpret_cast<const bf16x2*>(dv + o * D + 2 * lane);
        // transpose of the pair rotation: out0 = a0*c0 - a1*s0, out1 = a1*c1 + a0*s1
        dq0 = (float)ga[0] * c0 + (float)ga[1] * s1; dq1 = (float)ga[1] * c1 - (float)ga[0] * s0;
        dk0 = (float)gc[0] * c0 + (float)gc[1] * s1; dk1 = (float)gc[1] * c1 - (float)gc[0] * s0;
      }
      const float rq = rsqrtf(wave_sum(q0 * q0 + q1 * q1) / (float)D + eps);
      const float rk = rsqrtf(wave_sum(k0 * k0 + k1 * k1) / (float)D + eps);
      const float nq0 = q0 * rq, nq1 = q1 * rq, nk0 = k0 * rk, nk1 = k1 * rk;
      gq0 += dq0 * (float)(bf16)nq0; gq1 += dq1 * (float)(bf16)nq1; gk0 += dk0 * (float)(bf16)nk0; gk1 += dk1 * (float)(bf16)nk1;
      const float eq0 = dq0 * wq0, eq1 = dq1 * wq1, ek0 = dk0 * wk0, ek1 = dk1 * wk1;       // d(normalised)
      const float mq = wave_sum(eq0 * nq0 + eq1 * nq1) / (float)D, mk = wave_sum(ek0 * nk0 + ek1 * nk1) / (float)D;
      if (live) {
        const bf16x2 oq = {(bf16)(rq * (eq0 - nq0 * mq)), (bf16)(rq * (eq1 - nq1 * mq))};
        const bf16x2 ok = {(bf16)(rk * (ek0 - nk0 * mk)), (bf16)(rk * (ek1 - nk1 * mk))};
        *reinterpret_cast<bf16x2*>(dbase + (size_t)h * D + 2 * lane) = oq;
        *reinterpret_cast<bf16x2*>(dbase + (size_t)(H + h) * D + 2 * lane) = ok;
        *reinterpret_cast<bf16x2*>(dbase + (size_t)(2 * H + h) * D + 2 * lane) = dvv;
      }
    }
  }
  red[wave][0][lane] = gq0; red[wave][1][lane] = gq1; red[wave][2][lane] = gk0; red[wave][3][lane] = gk1;
  __syncthreads();
  if (threadIdx.x < 2 * D) {            // part[blk][0][d] = dq_weight, part[blk][1][d] = dk_weight
    const int which = threadIdx.x / D, d = threadIdx.x - which * D;
    const int slot = which * 2 + (d & 1), ln = d >> 1;
    part[((size_t)blockIdx.x * 2 + which) * D + d] = (red[0][slot][ln] + red[1][slot][ln]) + (red[2][slot][ln] + red[3][slot][ln]);
  }
}
// part[nblk][2][D] -> o0[d], o1[d]: 16 columns per block, 16 row groups per column (fixed assignment and a fixed-order LDS combine: deterministic)
__global__ __launch_bounds__(256) void colsum2_kernel(const float* __restrict__ part, float* __restrict__ o0, float* __restrict__ o1, int nblk, int D, int accumulate) {
  __shared__ float red[16][17];
  const int col = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + col;
  float a = 0.f;
  if (i < 2 * D) {
#pragma unroll 8
    for (int b = grp; b < nblk; b += 16) a += part[(size_t)b * 2 * D + i];   // independent loads: eight in flight per thread (was one: 34 us for 1.2 MB)
  }
  red[grp][col] = a;
  __syncthreads();
  if (threadIdx.x < 16 && i < 2 * D) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 16; g++) t += red[g][col];
    const int which = i / D, d = i - which * D;
    float* o = which ? o1 : o0;
    o[d] = (accumulate ? o[d] : 0.f) + t;
  }
}

static inline int grid_for(size_t n, int block = 256, int cap = 4096) {
  size_t g = (n + block - 1) / block;
  return (int)(g > (size_t)cap ? cap : (g < 1 ? 1 : g));
}

}  // namespace dmvae_dit
using namespace dmvae_dit;

// The eight-channel kernels whenever the width allows -- chosen by the WIDTH only, never by the row count: a 2B-sample call must equal two B-sample calls bit for bit
// (train.DMDTrainer's batched cond / uncond evaluation), and the two forms sum a row's squares in different lane orders.  46.9 vs 51.1 us at batch 64, 15.6 vs 14.7 at
// batch 16 (profiles/r5_dit_norm_passes.txt).  DMVAE_RM8=0: the four-channel kernels everywhere (A/B runs)
static bool rm8_on() { static const bool v = [] { const char* e = getenv("DMVAE_RM8"); return !(e && e[0] == '0'); }(); return v; }

extern "C" int dmvae_rmsnorm_modulate_bf16(const void* x, const void* w, const void* mod, void* y, int rows, int rows_per_sample, int c, int mod_stride,
                                           int shift_off, int scale_off, float eps, hipStream_t stream) {
  DMVAE_CHECK_ARG(x && w && mod && y && rows > 0 && rows_per_sample > 0, "rmsnorm_modulate_bf16: bad argument");
  DMVAE_CHECK_ARG(c % 4 == 0 && c >= 4 && c <= MAX_SWEEPS * 256, "rmsnorm_modulate_bf16: width must be a multiple of 4 up to 2048 (got %d)", c);
  DMVAE_CHECK_ARG(scale_off >= 0 && scale_off % 4 == 0 && (shift_off < 0 || shift_off % 4 == 0) && mod_stride % 4 == 0 && scale_off + c <= mod_stride,
                  "rmsnorm_modulate_bf16: modulation offsets must be multiples of 4 inside the row");
  if (c % 8 == 0 && mod_stride % 8 == 0 && scale_off % 8 == 0 && (shift_off < 0 || shift_off % 8 == 0) && rm8_on()) {
    switch ((c + 511) / 512) {
      case 1: hipLaunchKernelGGL((rmsnorm_modulate8_kernel<false, 1>), dim3((rows + 3) / 4), dim3(256), 0, stream, (float*)const_cast<void*>(x), (const float*)w,
                     (const bf16*)mod, (bf16*)y, rows, rows_per_sample, c, mod_stride, shift_off, scale_off, eps, (const bf16*)nullptr, (const bf16*)nullptr, 0, 0, (float*)nullptr); break;
      case 2: hipLaunchKernelGGL((rmsnorm_modulate8_kernel<false, 2>), dim3((rows + 3) / 4), dim3(256), 0, stream, (float*)const_cast<void*>(x), (const float*)w,
                     (const bf16*)mod, (bf16*)y, rows, rows_per_sample, c, mod_stride, shift_off, scale_off, eps, (const bf16*)nullptr, (const bf16*)nullptr, 0, 0, (float*)nullptr); break;
      case 3: hipLaunchKernelGGL((rmsnorm_modulate8_kernel<false, 3>), dim3((rows + 3) / 4), dim3(256), 0, stream, (float*)const_cast<void*>(x), (const float*)w,
                     (const bf16*)mod, (bf16*)y, rows, rows_per_sample, c, mod_stride, shift_off, scale_off, eps, (const bf16*)nullptr, (const bf16*)nullptr, 0, 0, (float*)nullptr); break;
      default: hipLaunchKernelGGL((rmsnorm_modulate8_kernel<false, 4>), dim3((rows + 3) / 4), dim3(256), 0, stream, (float*)const_cast<void*>(x), (const float*)w,
                     (const bf16*)mod, (bf16*)y, rows, rows_per_sample, c, mod_stride, shift_off, scale_off, eps, (const bf16*)nullptr, (const bf16*)nullptr, 0, 0, (float*)nullptr); break;
    }
  } else
  switch ((c + 255) / 256) {
    case 1: hipLaunchKernelGGL((rmsnorm_modulate_kernel<false, 1>), dim3((rows + 3) / 4), dim3(256), 0, stream, (float*)const_cast<void*>(x), (const float*)w,
                     (const bf16*)mod, (bf16*)y, rows, rows_per_sample, c, mod_stride, shift_off, scale_off, eps, (const bf16*)nullptr, (const bf16*)nullptr, 0, 0, (float*)nullptr); break;
    case 2: hipLaunchKernelGGL((rmsnorm_modulate_kernel<false, 2>), dim3((rows + 3) / 4), dim3(256), 0, stream, (float*)const_cast<void*>(x), (const float*)w,
                     (const bf16*)mod, (bf16*)y, rows, rows_per_sample, c, mod_stride, shift_off, scale_off, eps, (const bf16*)nullptr, (const bf16*)nullptr, 0, 0, (float*)nullptr); break;
    case 3: hipLaunchKernelGGL((rmsnorm_modulate_kernel<false, 3>), dim3((rows + 3) / 4), dim3(256), 0, stream, (float*)const_cast<void*>(x), (const float*)w,
                     (const bf16*)mod, (bf16*)y, rows, rows_per_sample, c, mod_stride, shift_off, scale_off, eps, (const bf16*)nullptr, (const bf16*)nullptr, 0, 0, (float*)nullptr); break;
    case 4: hipLaunchKernelGGL((rmsnorm_modulate_kernel<false, 4>), dim3((rows + 3) / 4), dim3(256), 0, stream, (float*)const_cast<void*>(x), (const float*)w,
                     (const bf16*)mod, (bf16*)y, rows, rows_per_sample, c, mod_stride, shift_off, scale_off, eps, (const bf16*)nullptr, (const bf16*)nullptr, 0, 0, (float*)nullptr); break;
    case 5: hipLaunchKernelGGL((rmsnorm_modulate_kernel<false, 5>), dim3((rows + 3) / 4), dim3(256), 0, stream, (float*)const_cast<void*>(x), (const float*)w,
                     (const bf16*)mod, (bf16*)y, rows, rows_per_sample, c, mod_stride, shift_off, scale_off, eps, (const bf16*)nullptr, (const bf16*)nullptr, 0, 0, (float*)nullptr); break;
    case 6: hipLaunchKernelGGL((rmsnorm_modulate_kernel<false, 6>), dim3((rows + 3) / 4), dim3(256), 0, stream, (float*)const_cast<void*>(x), (const float*)w,
                     (const bf16*)mod, (bf16*)y, rows, rows_per_sample, c, mod_stride, shift_off, scale_off, eps, (const bf16*)nullptr, (const bf16*)nullptr, 0, 0, (float*)nullptr); break;
    default: hipLaunchKernelGGL((rmsnorm_modulate_kernel<false, 8>), dim3((rows + 3) / 4), dim3(256), 0, stream, (float*)const_cast<void*>(x), (const float*)w,
                     (const bf16*)mod, (bf16*)y, rows, rows_per_sample, c, mod_stride, shift_off, scale_off, eps, (const bf16*)nullptr, (const bf16*)nullptr, 0, 0, (float*)nullptr); break;
  }
  DMVAE_CHECK_LAUNCH();
  return 0;
}

extern "C" int dmvae_gated_residual_rmsnorm_modulate(void* x, const void* r, const void* gate_mod, int gate_stride, int gate_off, const void* w,
                                                     const void* mod, void* y, int rows, int rows_per_sample, int c, int mod_stride, int shift_off,
                                                     int scale_off, float eps, hipStream_t stream) {
  DMVAE_CHECK_ARG(x && r && gate_mod && w && mod && y && rows > 0 && rows_per_sample > 0, "gated_residual_rmsnorm_modulate: bad argument");
  DMVAE_CHECK_ARG(c % 4 == 0 && c >= 4 && c <= MAX_SWEEPS * 256, "gated_residual_rmsnorm_modulate: width must be a multiple of 4 up to 2048 (got %d)", c);
  DMVAE_CHECK_ARG(scale_off >= 0 && scale_off % 4 == 0 && (shift_off < 0 || shift_off % 4 == 0) && mod_stride % 4 == 0 && scale_off + c <= mod_stride &&
                      gate_off >= 0 && gate_off % 4 == 0 && gate_stride % 4 == 0 && gate_off + c <= gate_stride,
                  "gated_residual_rmsnorm_modulate: modulation offsets must be multiples of 4 inside the row");
  if (c % 8 == 0 && mod_stride % 8 == 0 && scale_off % 8 == 0 && (shift_off < 0 || shift_off % 8 == 0) && gate_off % 8 == 0 && gate_stride % 8 == 0 && rm8_on()) {
    switch ((c + 511) / 512) {
      case 1: hipLaunchKernelGGL((rmsnorm_modulate8_kernel<true, 1>), dim3((rows + 3) / 4), dim3(256), 0, stream, (const float*)x, (const float*)w, (const bf16*)mod, (bf16*)y, rows,
                     rows_per_sample, c, mod_stride, shift_off, scale_off, eps, (const bf16*)r, (const bf16*)gate_mod, gate_stride, gate_off, (float*)x); break;
      case 2: hipLaunchKernelGGL((rmsnorm_modulate8_kernel<true, 2>), dim3((rows + 3) / 4), dim3(256), 0, stream, (const float*)x, (const float*)w, (const bf16*)mod, (bf16*)y, rows,
                     rows_per_sample, c, mod_stride, shift_off, scale_off, eps, (const bf16*)r, (const bf16*)gate_mod, gate_stride, gate_off, (float*)x); break;
      case 3: hipLaunchKernelGGL((rmsnorm_modulate8_kernel<true, 3>), dim3((rows + 3) / 4), dim3(256), 0, stream, (const float*)x, (const float*)w, (const bf16*)mod, (bf16*)y, rows,
                     rows_per_sample, c, mod_stride, shift_off, scale_off, eps, (const bf16*)r, (const bf16*)gate_mod, gate_stride, gate_off, (float*)x); break;
      default: hipLaunchKernelGGL((rmsnorm_modulate8_kernel<true, 4>), dim3((rows + 3) / 4), dim3(256), 0, stream, (const float*)x, (const float*)w, (const bf16*)mod, (bf16*)y, rows,
                     rows_per_sample, c, mod_stride, shift_off, scale_off, eps, (const bf16*)r, (const bf16*)gate_mod, gate_stride, gate_off, (float*)x); break;
    }
  } else
  switch ((c + 255) / 256) {
    case 1: hipLaunchKernelGGL((rmsnorm_modulate_kernel<true, 1>), dim3((rows + 3) / 4), dim3(256), 0, stream, (const float*)x, (const float*)w, (const bf16*)mod, (bf16*)y, rows,
                     rows_per_sample, c, mod_stride, shift_off, scale_off, eps, (const bf16*)r, (const bf16*)gate_mod, gate_stride, gate_off, (float*)x); break;
    case 2: hipLaunchKernelGGL((rmsnorm_modulate_kernel<true, 2>), dim3((rows + 3) / 4), dim3(256), 0, stream, (const float*)x, (const float*)w, (const bf16*)mod, (bf16*)y, rows,
                     rows_per_sample, c, mod_stride, shift_off, scale_off, eps, (const bf16*)r, (const bf16*)gate_mod, gate_stride, gate_off, (float*)x); break;
    case 3: hipLaunchKernelGGL((rmsnorm_modulate_kernel<true, 3>), dim3((rows + 3) / 4), dim3(256), 0, stream, (const float*)x, (const float*)w, (const bf16*)mod, (bf16*)y, rows,
                     rows_per_sample, c, mod_stride, shift_off, scale_off, eps, (const bf16*)r, (const bf16*)gate_mod, gate_stride, gate_off, (float*)x); break;
    case 4: hipLaunchKernelGGL((rmsnorm_modulate_kernel<true, 4>), dim3((rows + 3) / 4), dim3(256), 0, stream, (const float*)x, (const float*)w, (const bf16*)mod, (bf16*)y, rows,
                     rows_per_sample, c, mod_stride, shift_off, scale_off, eps, (const bf16*)r, (const bf16*)gate_mod, gate_stride, gate_off, (float*)x); break;
    case 5: hipLaunchKernelGGL((rmsnorm_modulate_kernel<true, 5>), dim3((rows + 3) / 4), dim3(256), 0, stream, (const float*)x, (const float*)w, (const bf16*)mod, (bf16*)y, rows,
                     rows_per_sample, c, mod_stride, shift_off, scale_off, eps, (const bf16*)r, (const bf16*)gate_mod, gate_stride, gate_off, (float*)x); break;
    case 6: hipLaunchKernelGGL((rmsnorm_modulate_kernel<true, 6>), dim3((rows + 3) / 4), dim3(256), 0, stream, (const float*)x, (const float*)w, (const bf16*)mod, (bf16*)y, rows,
                     rows_per_sample, c, mod_stride, shift_off, scale_off, eps, (const bf16*)r, (const bf16*)gate_mod, gate_stride, gate_off, (float*)x); break;
    default: hipLaunchKernelGGL((rmsnorm_modulate_kernel<true, 8>), dim3((rows + 3) / 4), dim3(256), 0, stream, (const float*)x, (const float*)w, (const bf16*)mod, (bf16*)y, rows,
                     rows_per_sample, c, mod_stride, shift_off, scale_off, eps, (const bf16*)r, (const bf16*)gate_mod, gate_stride, gate_off, (float*)x); break;
  }
  DMVAE_CHECK_LAUNCH();
  return 0;
}

// Out-of-place form for the training route, where the residual stream before the update is kept for the backward pass: x_out = x_in + bf16(gate * r), and,
// when y is given, y = rmsnorm_modulate(x_out) in the same pass (w, mod, offsets then required).  Saves the clone the in-place kernels need there.
extern "C" int dmvae_gated_residual_out(const void* x_in, void* x_out, const void* r, const void* gate_mod, int gate_stride, int gate_off, const void* w,
                                        const void* mod, void* y, int rows, int rows_per_sample, int c, int mod_stride, int shift_off, int scale_off,
                                        float eps, hipStream_t stream) {
  DMVAE_CHECK_ARG(x_in && x_out && r && gate_mod && rows > 0 && rows_per_sample > 0 && (!y || (w && mod)), "gated_residual_out: bad argument");
  DMVAE_CHECK_ARG(c % 4 == 0 && c >= 4 && c <= MAX_SWEEPS * 256, "gated_residual_out: width must be a multiple of 4 up to 2048 (got %d)", c);
  DMVAE_CHECK_ARG(gate_off >= 0 && gate_off % 4 == 0 && gate_stride % 4 == 0 && gate_off + c <= gate_stride &&
                      (!y || (scale_off >= 0 && scale_off % 4 == 0 && (shift_off < 0 || shift_off % 4 == 0) && mod_stride % 4 == 0 && scale_off + c <= mod_stride)),
                  "gated_residual_out: modulation offsets must be multiples of 4 inside the row");
  if (c % 8 == 0 && mod_stride % 8 == 0 && scale_off % 8 == 0 && (shift_off < 0 || shift_off % 8 == 0) && gate_off % 8 == 0 && gate_stride % 8 == 0 && rm8_on()) {
    switch ((c + 511) / 512) {
      case 1: hipLaunchKernelGGL((rmsnorm_modulate8_kernel<true, 1>), dim3((rows + 3) / 4), dim3(256), 0, stream, (const float*)x_in, (const float*)w, (const bf16*)mod, (bf16*)y,
                     rows, rows_per_sample, c, mod_stride, shift_off, scale_off, eps, (const bf16*)r, (const bf16*)gate_mod, gate_stride, gate_off, (float*)x_out); break;
      case 2: hipLaunchKernelGGL((rmsnorm_modulate8_kernel<true, 2>), dim3((rows + 3) / 4), dim3(256), 0, stream, (const float*)x_in, (const float*)w, (const bf16*)mod, (bf16*)y,
                     rows, rows_per_sample, c, mod_stride, shift_off, scale_off, eps, (const bf16*)r, (const bf16*)gate_mod, gate_stride, gate_off, (float*)x_out); break;
      case 3: hipLaunchKernelGGL((rmsnorm_modulate8_kernel<true, 3>), dim3((rows + 3) / 4), dim3(256), 0, stream, (const float*)x_in, (const float*)w, (const bf16*)mod, (bf16*)y,
                     rows, rows_per_sample, c, mod_stride, shift_off, scale_off, eps, (const bf16*)r, (const bf16*)gate_mod, gate_stride, gate_off, (float*)x_out); break;
      default: hipLaunchKernelGGL((rmsnorm_modulate8_kernel<true, 4>), dim3((rows + 3) / 4), dim3(256), 0, stream, (const float*)x_in, (const float*)w, (const bf16*)mod, (bf16*)y,
                     rows, rows_per_sample, c, mod_stride, shift_off, scale_off, eps, (const bf16*)r, (const bf16*)gate_mod, gate_stride, gate_off, (float*)x_out); break;
    }
  } else
  switch ((c + 255) / 256) {
    case 1: hipLaunchKernelGGL((rmsnorm_modulate_kernel<true, 1>), dim3((rows + 3) / 4), dim3(256), 0, stream, (const float*)x_in, (const float*)w, (const bf16*)mod, (bf16*)y,
                     rows, rows_per_sample, c, mod_stride, shift_off, scale_off, eps, (const bf16*)r, (const bf16*)gate_mod, gate_stride, gate_off, (float*)x_out); break;
    case 2: hipLaunchKernelGGL((rmsnorm_modulate_kernel<true, 2>), dim3((rows + 3) / 4), dim3(256), 0, stream, (const float*)x_in, (const float*)w, (const bf16*)mod, (bf16*)y,
                     rows, rows_per_sample, c, mod_stride, shift_off, scale_off, eps, (const bf16*)r, (const bf16*)gate_mod, gate_stride, gate_off, (float*)x_out); break;
    case 3: hipLaunchKernelGGL((rmsnorm_modulate_kernel<true, 3>), dim3((rows + 3) / 4), dim3(256), 0, stream, (const float*)x_in, (const float*)w, (const bf16*)mod, (bf16*)y,
                     rows, rows_per_sample, c, mod_stride, shift_off, scale_off, eps, (const bf16*)r, (const bf16*)gate_mod, gate_stride, gate_off, (float*)x_out); break;
    case 4: hipLaunchKernelGGL((rmsnorm_modulate_kernel<true, 4>), dim3((rows + 3) / 4), dim3(256), 0, stream, (const float*)x_in, (const float*)w, (const bf16*)mod, (bf16*)y,
                     rows, rows_per_sample, c, mod_stride, shift_off, scale_off, eps, (const bf16*)r, (const bf16*)gate_mod, gate_stride, gate_off, (float*)x_out); break;
    case 5: hipLaunchKernelGGL((rmsnorm_modulate_kernel<true, 5>), dim3((rows + 3) / 4), dim3(256), 0, stream, (const float*)x_in, (const float*)w, (const bf16*)mod, (bf16*)y,
                     rows, rows_per_sample, c, mod_stride, shift_off, scale_off, eps, (const bf16*)r, (const bf16*)gate_mod, gate_stride, gate_off, (float*)x_out); break;
    case 6: hipLaunchKernelGGL((rmsnorm_modulate_kernel<true, 6>), dim3((rows + 3) / 4), dim3(256), 0, stream, (const float*)x_in, (const float*)w, (const bf16*)mod, (bf16*)y,
                     rows, rows_per_sample, c, mod_stride, shift_off, scale_off, eps, (const bf16*)r, (const bf16*)gate_mod, gate_stride, gate_off, (float*)x_out); break;
    default: hipLaunchKernelGGL((rmsnorm_modulate_kernel<true, 8>), dim3((rows + 3) / 4), dim3(256), 0, stream, (const float*)x_in, (const float*)w, (const bf16*)mod, (bf16*)y,
                     rows, rows_per_sample, c, mod_stride, shift_off, scale_off, eps, (const bf16*)r, (const bf16*)gate_mod, gate_stride, gate_off, (float*)x_out); break;
  }
  DMVAE_CHECK_LAUNCH();
  return 0;
}

namespace dmvae_dit {
// 16 lanes per (token, head) row, 8 channels (16 B) per lane: the same arithmetic as qknorm_rope_kernel with a quarter of the memory instructions
// (head dims that are multiples of 8: 64 -> 8 live lanes of 16, 72 -> 9).
__global__ __launch_bounds__(256) void qknorm_rope16_kernel(const bf16* __restrict__ qkv, const float* __restrict__ qw, const float* __restrict__ kw,
                                                            const float* __restrict__ cosb, const float* __restrict__ sinb, bf16* __restrict__ qo,
                                                            bf16* __restrict__ ko, bf16* __restrict__ vo, int tokens, int N, int H, int D, int Dp,
                                                            float eps) {
  const int sub = threadIdx.x & 15;
  const int row = blockIdx.x * 16 + (threadIdx.x >> 4);
  const bool inr = row < tokens * H;
  const int tok = inr ? row / H : 0, h = inr ? row - tok * H : 0;
  const int b = tok / N, n = tok - b * N;
  const int d0 = sub * 8;
  const bool live = inr && d0 < D;
  uint4 q = {0, 0, 0, 0}, k = {0, 0, 0, 0}, v = {0, 0, 0, 0};
  if (live) {
    const bf16* base = qkv + (size_t)tok * 3 * H * D + d0;
    q = *reinterpret_cast<const uint4*>(base + (size_t)h * D);
    k = *reinterpret_cast<const uint4*>(base + (size_t)(H + h) * D);
    v = *reinterpret_cast<const uint4*>(base + (size_t)(2 * H + h) * D);
  }
  float sq = dmvae_sumsq8(q), sk = dmvae_sumsq8(k);
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) { sq += __shfl_xor(sq, o, 64); sk += __shfl_xor(sk, o, 64); }
  if (!inr || d0 >= Dp) return;
  const size_t o = ((size_t)(b * H + h) * N + n);
  if (live) {
    q = dmvae_norm_rope8(q, rsqrtf(sq / (float)D + eps), qw, cosb, sinb, n, D, d0);
    k = dmvae_norm_rope8(k, rsqrtf(sk / (float)D + eps), kw, cosb, sinb, n, D, d0);
    *reinterpret_cast<uint4*>(vo + o * D + d0) = v;
  }
  *reinterpret_cast<uint4*>(qo + o * Dp + d0) = q;      // zeros in the padding chunks
  *reinterpret_cast<uint4*>(ko + o * Dp + d0) = k;
}
}  // namespace dmvae_dit

extern "C" int dmvae_qknorm_rope_bf16(const void* qkv, const void* q_weight, const void* k_weight, const void* cos_table, const void* sin_table,
                                      void* q_out, void* k_out, void* v_out, int batch, int seq, int heads, int head_dim, int head_dim_padded,
                                      float eps, hipStream_t stream) {
  DMVAE_CHECK_ARG(qkv && q_weight && k_weight && cos_table && sin_table && q_out && k_out && v_out && batch > 0 && seq > 0 && heads > 0,
                  "qknorm_rope_bf16: bad argument");
  DMVAE_CHECK_ARG(head_dim % 2 == 0 && head_dim >= 2 && head_dim_padded >= head_dim && head_dim_padded % 2 == 0 && head_dim_padded <= 128,
                  "qknorm_rope_bf16: head dim must be even, padded head dim <= 128 (got %d, %d)", head_dim, head_dim_padded);
  const int tokens = batch * seq;
  if (head_dim % 8 == 0 && head_dim_padded % 8 == 0) {
    hipLaunchKernelGGL(qknorm_rope16_kernel, dim3((tokens * heads + 15) / 16), dim3(256), 0, stream, (const bf16*)qkv, (const float*)q_weight,
                       (const float*)k_weight, (const float*)cos_table, (const float*)sin_table, (bf16*)q_out, (bf16*)k_out, (bf16*)v_out, tokens, seq, heads,
                       head_dim, head_dim_padded, eps);
    DMVAE_CHECK_LAUNCH();
    return 0;
  }
  hipLaunchKernelGGL(qknorm_rope_kernel, dim3((tokens * heads + 3) / 4), dim3(256), 0, stream, (const bf16*)qkv, (const float*)q_weight, (const float*)k_weight,
                     (const float*)cos_table, (const float*)sin_table, (bf16*)q_out, (bf16*)k_out, (bf16*)v_out, tokens, seq, heads, head_dim,
                     head_dim_padded, eps);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

extern "C" int dmvae_swiglu_bf16(const void* x12, void* out, size_t rows, int hidden, hipStream_t stream) {
  DMVAE_CHECK_ARG(x12 && out && hidden > 0 && hidden % 8 == 0, "swiglu_bf16: hidden width must be a multiple of 8 (got %d)", hidden);
  if (rows == 0) return 0;
  hipLaunchKernelGGL(swiglu_kernel, dim3(grid_for(rows * (size_t)(hidden / 8))), dim3(256), 0, stream, (const bf16*)x12, (bf16*)out, rows, hidden / 8);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

extern "C" int dmvae_gated_residual_f32(void* x, const void* y, const void* mod, size_t rows, int rows_per_sample, int c, int mod_stride, int gate_off,
                                        hipStream_t stream) {
  DMVAE_CHECK_ARG(x && y && mod && rows_per_sample > 0 && c > 0 && c % 8 == 0 && gate_off >= 0 && gate_off % 8 == 0 && mod_stride % 8 == 0 &&
                      gate_off + c <= mod_stride, "gated_residual_f32: width / offsets must be multiples of 8 inside the modulation row");
  if (rows == 0) return 0;
  hipLaunchKernelGGL(gated_residual_kernel, dim3(grid_for(rows * (size_t)(c / 8))), dim3(256), 0, stream, (float*)x, (const bf16*)y, (const bf16*)mod, rows,
                     c / 8, rows_per_sample, mod_stride, gate_off);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

// ---- backward entry points -----------------------------------------------------------------------------------------------------------
extern "C" size_t dmvae_dit_bwd_workspace(int batch, int c) {
  const size_t a = ((size_t)batch * RM_BPS * 3 + batch) * (size_t)c * sizeof(float) + (size_t)batch * RM_MAX_SEQ * sizeof(float2),
               b = (size_t)2048 * 2 * 128 * sizeof(float);
  return a > b ? a : b;
}

extern "C" int dmvae_gated_residual_bwd(const void* dx, const void* y, const void* mod, void* dy, void* dmod, int batch, int seq, int c, int mod_stride,
                                        int gate_off, hipStream_t stream) {
  DMVAE_CHECK_ARG(dx && y && mod && dy && dmod && batch > 0 && seq > 0 && c > 0 && c % 8 == 0 && gate_off >= 0 && gate_off % 8 == 0 &&
                      mod_stride % 8 == 0 && gate_off + c <= mod_stride, "gated_residual_bwd: bad argument");
  hipLaunchKernelGGL(gated_residual_bwd_kernel, dim3((c + 255) / 256, batch), dim3(256), 0, stream, (const float*)dx, (const bf16*)y, (const bf16*)mod,
                     (bf16*)dy, (float*)dmod, seq, c, mod_stride, gate_off);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

extern "C" int dmvae_swiglu_bwd(const void* dh, const void* x12, void* dx12, size_t rows, int hidden, hipStream_t stream) {
  DMVAE_CHECK_ARG(dh && x12 && dx12 && hidden > 0 && hidden % 8 == 0, "swiglu_bwd: hidden width must be a multiple of 8 (got %d)", hidden);
  if (rows == 0) return 0;
  hipLaunchKernelGGL(swiglu_bwd_kernel, dim3(grid_for(rows * (size_t)(hidden / 8))), dim3(256), 0, stream, (const bf16*)dh, (const bf16*)x12, (bf16*)dx12,
                     rows, hidden / 8);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

extern "C" int dmvae_rmsnorm_modulate_bwd(const void* da, const void* x, const void* w, const void* mod, void* dx_io, void* dmod, void* dw,
                                          void* workspace, size_t workspace_bytes, int batch, int seq, int c, int mod_stride, int shift_off,
                                          int scale_off, float eps, int accumulate, hipStream_t stream) {
  DMVAE_CHECK_ARG(da && x && w && mod && dx_io && dmod && workspace && batch > 0 && seq > 0, "rmsnorm_modulate_bwd: bad argument");
  DMVAE_CHECK_ARG(c % 4 == 0 && c >= 4 && c <= MAX_SWEEPS * 256, "rmsnorm_modulate_bwd: width must be a multiple of 4 up to 2048 (got %d)", c);
  DMVAE_CHECK_ARG(scale_off >= 0 && scale_off % 4 == 0 && (shift_off < 0 || shift_off % 4 == 0) && mod_stride % 4 == 0 && scale_off + c <= mod_stride,
                  "rmsnorm_modulate_bwd: modulation offsets must be multiples of 4 inside the row");
  DMVAE_CHECK_ARG(workspace_bytes >= dmvae_dit_bwd_workspace(batch, c), "rmsnorm_modulate_bwd: workspace too small");
  const size_t lds = (size_t)2 * 3 * c * sizeof(float);
  // blocks per sample: enough blocks to fill the chip (~1024), few enough that the per-block partial sums (3 * c floats through LDS and the workspace) stay small
  // next to the rows a block walks -- 32 at batch 16 (2 rows per wave at 256 tokens), 16 at batch 64
  int bps = 1024 / batch;
  bps = bps > RM_BPS ? RM_BPS : (bps < 4 ? 4 : bps);
  constexpr bool split_ok = true;
  if (split_ok && seq <= RM_MAX_SEQ) {
    float2* rowstat = reinterpret_cast<float2*>((float*)workspace + ((size_t)batch * RM_BPS * 3 + batch) * (size_t)c);
    const int rows = batch * seq;
    hipLaunchKernelGGL(rmsnorm_rowstat_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, (const bf16*)da, (const float*)x, (const float*)w, (const bf16*)mod,
                       rowstat, rows, seq, c, mod_stride, scale_off, eps);
    DMVAE_CHECK_LAUNCH();
    const int bt = ((c / 4) + 63) / 64 * 64;
    hipLaunchKernelGGL(rmsnorm_modulate_bwd_apply_kernel, dim3(bps, batch), dim3(bt), 0, stream, (const bf16*)da, (const float*)x, (const float*)w,
                       (const bf16*)mod, rowstat, (float*)dx_io, (float*)workspace, seq, c, mod_stride, scale_off);
    DMVAE_CHECK_LAUNCH();
  } else {
  auto go = [&](auto sw) {
    constexpr int SW = decltype(sw)::value;
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rmsnorm_modulate_bwd_kernel<SW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_lds = lds;
    }
    hipLaunchKernelGGL(rmsnorm_modulate_bwd_kernel<SW>, dim3(bps, batch), dim3(256), lds, stream, (const bf16*)da, (const float*)x, (const float*)w,
                       (const bf16*)mod, (float*)dx_io, (float*)workspace, seq, c, mod_stride, scale_off, eps);
  };
  switch ((c + 255) / 256) {
    case 1: go(std::integral_constant<int, 1>{}); break;
    case 2: go(std::integral_constant<int, 2>{}); break;
    case 3: go(std::integral_constant<int, 3>{}); break;
    case 4: go(std::integral_constant<int, 4>{}); break;
    case 5: go(std::integral_constant<int, 5>{}); break;
    case 6: go(std::integral_constant<int, 6>{}); break;
    default: go(std::integral_constant<int, 8>{}); break;
  }
  DMVAE_CHECK_LAUNCH();
  }
  float* wpart = (float*)workspace + (size_t)batch * RM_BPS * 3 * c;
  hipLaunchKernelGGL(rmsnorm_modulate_bwd_final_kernel, dim3((c + 255) / 256, batch), dim3(256), 0, stream, (const float*)workspace, (float*)dmod, wpart,
                     bps, c, mod_stride, shift_off, scale_off);
  DMVAE_CHECK_LAUNCH();
  if (dw) {
    hipLaunchKernelGGL(rmsnorm_modulate_bwd_weight_kernel, dim3((c + 255) / 256), dim3(256), 0, stream, wpart, (float*)dw, batch, c, accumulate);
    DMVAE_CHECK_LAUNCH();
  }
  return 0;
}

namespace dmvae_dit {
// Backward with the same 16-lanes-per-row mapping (head dim % 8 == 0): per lane 8 channels of q and k, the two per-row reductions as 16-lane butterflies,
// weight-gradient accumulators per lane over the rows it walks, combined over the block's 16 row groups in a fixed order.
__global__ __launch_bounds__(256) void qknorm_rope16_bwd_kernel(const bf16* __restrict__ dq, const bf16* __restrict__ dk, const bf16* __restrict__ dv,
                                                                const bf16* __restrict__ qkv, const float* __restrict__ qw, const float* __restrict__ kw,
                                                                const float* __restrict__ cosb, const float* __restrict__ sinb, bf16* __restrict__ dqkv,
                                                                float* __restrict__ part, int tokens, int N, int H, int D, int Dp, float eps) {
  __shared__ float red[16][16][17];
  const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int d0 = sub * 8;
  const bool lane_live = d0 < D;
  float wq[8], wk[8], gq[8], gk[8];
#pragma unroll
  for (int e = 0; e < 8; e++) { wq[e] = lane_live ? qw[d0 + e] : 0.f; wk[e] = lane_live ? kw[d0 + e] : 0.f; gq[e] = 0.f; gk[e] = 0.f; }
  const int rows = tokens * H;
  for (int row0 = blockIdx.x * 16; row0 < rows; row0 += gridDim.x * 16) {
    const int row = row0 + grp;
    const bool live = lane_live && row < rows;
    const int tok = row < rows ? row / H : 0, h = row < rows ? row - tok * H : 0;
    const int b = tok / N, n = tok - b * N;
    const size_t o = ((size_t)(b * H + h) * N + n);
    float q[8], k[8], eq[8], ek[8], nq[8], nk[8];
    uint4 dvv = {0, 0, 0, 0};
    float sq = 0.f, sk = 0.f;
#pragma unroll
    for (int e = 0; e < 8; e++) { q[e] = 0.f; k[e] = 0.f; eq[e] = 0.f; ek[e] = 0.f; }
    if (live) {
      const bf16* base = qkv + (size_t)tok * 3 * H * D + d0;
      const uint4 qa = *reinterpret_cast<const uint4*>(base + (size_t)h * D), ka = *reinterpret_cast<const uint4*>(base + (size_t)(H + h) * D);
      const uint4 ga = *reinterpret_cast<const uint4*>(dq + o * Dp + d0), gc = *reinterpret_cast<const uint4*>(dk + o * Dp + d0);
      dvv = *reinterpret_cast<const uint4*>(dv + o * D + d0);
      const bf16x8 qb = *reinterpret_cast<const bf16x8*>(&qa), kb = *reinterpret_cast<const bf16x8*>(&ka);
      const bf16x8 gqb = *reinterpret_cast<const bf16x8*>(&ga), gkb = *reinterpret_cast<const bf16x8*>(&gc);
      const float4 c0 = *reinterpret_cast<const float4*>(cosb + (size_t)n * D + d0), c1 = *reinterpret_cast<const float4*>(cosb + (size_t)n * D + d0 + 4);
      const float4 s0 = *reinterpret_cast<const float4*>(sinb + (size_t)n * D + d0), s1 = *reinterpret_cast<const float4*>(sinb + (size_t)n * D + d0 + 4);
      const float cv[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w}, sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
      for (int e = 0; e < 8; e += 2) {  // transpose of the pair rotation: out0 = a0*c0 - a1*s0, out1 = a1*c1 + a0*s1
        q[e] = (float)qb[e]; q[e + 1] = (float)qb[e + 1]; k[e] = (float)kb[e]; k[e + 1] = (float)kb[e + 1];
        eq[e] = (float)gqb[e] * cv[e] + (float)gqb[e + 1] * sv[e + 1]; eq[e + 1] = (float)gqb[e + 1] * cv[e + 1] - (float)gqb[e] * sv[e];
        ek[e] = (float)gkb[e] * cv[e] + (float)gkb[e + 1] * sv[e + 1]; ek[e + 1] = (float)gkb[e + 1] * cv[e + 1] - (float)gkb[e] * sv[e];
      }
#pragma unroll
      for (int e = 0; e < 8; e++) { sq += q[e] * q[e]; sk += k[e] * k[e]; }
    }
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) { sq += __shfl_xor(sq, m, 64); sk += __shfl_xor(sk, m, 64); }
    const float rq = rsqrtf(sq / (float)D + eps), rk = rsqrtf(sk / (float)D + eps);
    float mq = 0.f, mk = 0.f;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      nq[e] = q[e] * rq; nk[e] = k[e] * rk;
      gq[e] += eq[e] * (float)(bf16)nq[e]; gk[e] += ek[e] * (float)(bf16)nk[e];       // d(weight): the weight multiplies the bf16-rounded normalised value
      eq[e] *= wq[e]; ek[e] *= wk[e];                                                 // d(normalised)
      mq += eq[e] * nq[e]; mk += ek[e] * nk[e];
    }
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) { mq += __shfl_xor(mq, m, 64); mk += __shfl_xor(mk, m, 64); }
    mq /= (float)D; mk /= (float)D;
    if (live) {
      uint4 oq, ok;
      oq.x = dmvae_pack_bf16x2(rq * (eq[0] - nq[0] * mq), rq * (eq[1] - nq[1] * mq)); oq.y = dmvae_pack_bf16x2(rq * (eq[2] - nq[2] * mq), rq * (eq[3] - nq[3] * mq));
      oq.z = dmvae_pack_bf16x2(rq * (eq[4] - nq[4] * mq), rq * (eq[5] - nq[5] * mq)); oq.w = dmvae_pack_bf16x2(rq * (eq[6] - nq[6] * mq), rq * (eq[7] - nq[7] * mq));
      ok.x = dmvae_pack_bf16x2(rk * (ek[0] - nk[0] * mk), rk * (ek[1] - nk[1] * mk)); ok.y = dmvae_pack_bf16x2(rk * (ek[2] - nk[2] * mk), rk * (ek[3] - nk[3] * mk));
      ok.z = dmvae_pack_bf16x2(rk * (ek[4] - nk[4] * mk), rk * (ek[5] - nk[5] * mk)); ok.w = dmvae_pack_bf16x2(rk * (ek[6] - nk[6] * mk), rk * (ek[7] - nk[7] * mk));
      bf16* dbase = dqkv + (size_t)tok * 3 * H * D + d0;
      *reinterpret_cast<uint4*>(dbase + (size_t)h * D) = oq;
      *reinterpret_cast<uint4*>(dbase + (size_t)(H + h) * D) = ok;
      *reinterpret_cast<uint4*>(dbase + (size_t)(2 * H + h) * D) = dvv;
    }
  }
#pragma unroll
  for (int e = 0; e < 8; e++) { red[grp][sub][e] = gq[e]; red[grp][sub][8 + e] = gk[e]; }
  __syncthreads();
  if (threadIdx.x < 2 * D) {            // part[blk][0][d] = dq_weight partial, part[blk][1][d] = dk_weight partial
    const int which = threadIdx.x / D, d = threadIdx.x - which * D;
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 16; g++) t += red[g][d >> 3][which * 8 + (d & 7)];
    part[((size_t)blockIdx.x * 2 + which) * D + d] = t;
  }
}
}  // namespace dmvae_dit

static int qk_bwd_nblk(int batch, int seq, int heads, int head_dim, int head_dim_padded) {
  const int tokens = batch * seq;
  int nblk = (tokens * heads + 3) / 4; if (nblk > 2048) nblk = 2048;   // eight blocks per CU: the rows are short (4-B lanes, four wave reductions each) and latency-bound
  if (head_dim % 8 == 0 && head_dim_padded % 8 == 0 && 2 * head_dim <= 256) {
    nblk = (tokens * heads + 15) / 16; if (nblk > 2048) nblk = 2048;
  }
  return nblk;
}
// blocks of the first stage = rows of its partial-sum array [nblk][2][head_dim] f32 (the workspace of dmvae_qknorm_rope_bwd; `part` of dmvae_qknorm_rope_bwd_partial)
extern "C" int dmvae_qknorm_rope_bwd_nblk(int batch, int seq, int heads, int head_dim, int head_dim_padded) {
  return qk_bwd_nblk(batch, seq, heads, head_dim, head_dim_padded);
}

static int qk_bwd_launch(const void* dq, const void* dk, const void* dv, const void* qkv, const void* q_weight, const void* k_weight, const void* cos_table,
                         const void* sin_table, void* dqkv, void* part, size_t part_bytes, int batch, int seq, int heads, int head_dim, int head_dim_padded, float eps,
                         hipStream_t stream, int* nblk_out) {
  DMVAE_CHECK_ARG(dq && dk && dv && qkv && q_weight && k_weight && cos_table && sin_table && dqkv && part && batch > 0 && seq > 0 && heads > 0,
                  "qknorm_rope_bwd: bad argument");
  DMVAE_CHECK_ARG(head_dim % 2 == 0 && head_dim >= 2 && head_dim_padded >= head_dim && head_dim_padded <= 128,
                  "qknorm_rope_bwd: head dim must be even, padded head dim <= 128 (got %d, %d)", head_dim, head_dim_padded);
  const int tokens = batch * seq;
  const int nblk = qk_bwd_nblk(batch, seq, heads, head_dim, head_dim_padded);
  DMVAE_CHECK_ARG(part_bytes >= (size_t)nblk * 2 * head_dim * sizeof(float), "qknorm_rope_bwd: workspace too small");
  if (head_dim % 8 == 0 && head_dim_padded % 8 == 0 && 2 * head_dim <= 256) {
    hipLaunchKernelGGL(qknorm_rope16_bwd_kernel, dim3(nblk), dim3(256), 0, stream, (const bf16*)dq, (const bf16*)dk, (const bf16*)dv, (const bf16*)qkv,
                       (const float*)q_weight, (const float*)k_weight, (const float*)cos_table, (const float*)sin_table, (bf16*)dqkv, (float*)part,
                       tokens, seq, heads, head_dim, head_dim_padded, eps);
  } else
  hipLaunchKernelGGL(qknorm_rope_bwd_kernel, dim3(nblk), dim3(256), 0, stream, (const bf16*)dq, (const bf16*)dk, (const bf16*)dv, (const bf16*)qkv,
                     (const float*)q_weight, (const float*)k_weight, (const float*)cos_table, (const float*)sin_table, (bf16*)dqkv, (float*)part,
                     tokens, seq, heads, head_dim, head_dim_padded, eps);
  DMVAE_CHECK_LAUNCH();
  *nblk_out = nblk;
  return 0;
}

extern "C" int dmvae_qknorm_rope_bwd(const void* dq, const void* dk, const void* dv, const void* qkv, const void* q_weight, const void* k_weight,
                                     const void* cos_table, const void* sin_table, void* dqkv, void* dq_weight, void* dk_weight, void* workspace,
                                     size_t workspace_bytes, int batch, int seq, int heads, int head_dim, int head_dim_padded, float eps, int accumulate,
                                     hipStream_t stream) {
  DMVAE_CHECK_ARG(dq_weight && dk_weight, "qknorm_rope_bwd: bad argument");
  int nblk = 0;
  const int rc = qk_bwd_launch(dq, dk, dv, qkv, q_weight, k_weight, cos_table, sin_table, dqkv, workspace, workspace_bytes, batch, seq, heads, head_dim, head_dim_padded, eps,
                               stream, &nblk);
  if (rc) return rc;
  hipLaunchKernelGGL(colsum2_kernel, dim3((2 * head_dim + 15) / 16), dim3(256), 0, stream, (const float*)workspace, (float*)dq_weight, (float*)dk_weight,
                     nblk, head_dim, accumulate);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

// First stage only: the per-block partial sums of the two norm-weight gradients stay in `part` ([dmvae_qknorm_rope_bwd_nblk][2][head_dim] f32) for a reduction the
// caller batches over layers (dmvae_colsum2_batched, csrc/dit_stack.hip).
extern "C" int dmvae_qknorm_rope_bwd_partial(const void* dq, const void* dk, const void* dv, const void* qkv, const void* q_weight, const void* k_weight,
                                             const void* cos_table, const void* sin_table, void* dqkv, void* part, size_t part_bytes, int batch, int seq, int heads,
                                             int head_dim, int head_dim_padded, float eps, hipStream_t stream) {
  int nblk = 0;
  return qk_bwd_launch(dq, dk, dv, qkv, q_weight, k_weight, cos_table, sin_table, dqkv, part, part_bytes, batch, seq, heads, head_dim, head_dim_padded, eps, stream, &nblk);
}
