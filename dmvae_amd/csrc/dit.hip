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
#include "dmvae_hip.h"

namespace dmvae_dit {

constexpr int MAX_SWEEPS = 8;  // C <= 2048

// one wave per row; mod: [B][stride] bf16 (the adaLN Linear's output), shift_off < 0: no shift
__global__ __launch_bounds__(256) void rmsnorm_modulate_kernel(const float* __restrict__ x, const float* __restrict__ w, const bf16* __restrict__ mod,
                                                               bf16* __restrict__ y, int rows, int rows_per_sample, int C, int stride, int shift_off,
                                                               int scale_off, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (size_t)row * C;
  f32x4 v[MAX_SWEEPS];
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < MAX_SWEEPS; k++) {
    const int c = k * 256 + lane * 4;
    if (c < C) {
      v[k] = *reinterpret_cast<const f32x4*>(xr + c);
      ss += (v[k][0] * v[k][0] + v[k][1] * v[k][1]) + (v[k][2] * v[k][2] + v[k][3] * v[k][3]);
    }
  }
  const float rs = rsqrtf(wave_sum(ss) / (float)C + eps);
  const bf16* mrow = mod + (size_t)(row / rows_per_sample) * stride;
  bf16* yr = y + (size_t)row * C;
#pragma unroll
  for (int k = 0; k < MAX_SWEEPS; k++) {
    const int c = k * 256 + lane * 4;
    if (c < C) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(w + c);
      const bf16x4 sc = *reinterpret_cast<const bf16x4*>(mrow + scale_off + c);
      bf16x4 sh = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
      if (shift_off >= 0) sh = *reinterpret_cast<const bf16x4*>(mrow + shift_off + c);
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; e++)  // `1 + scale` is a bf16 tensor in the reference's autocast graph (bf16 scale): rounded before it multiplies
        o[e] = (bf16)(v[k][e] * rs * g[e] * (float)(bf16)(1.f + (float)sc[e]) + (float)sh[e]);
      *reinterpret_cast<bf16x4*>(yr + c) = o;
    }
  }
}

// one wave per token (b, n); lane j < D/2 owns the feature pair (2j, 2j+1) of every head in turn
__global__ __launch_bounds__(256) void qknorm_rope_kernel(const bf16* __restrict__ qkv, const float* __restrict__ qw, const float* __restrict__ kw,
                                                          const float* __restrict__ cosb, const float* __restrict__ sinb, bf16* __restrict__ qo,
                                                          bf16* __restrict__ ko, bf16* __restrict__ vo, int tokens, int N, int H, int D, int Dp,
                                                          float eps) {
  const int tok = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (tok >= tokens) return;
  const int b = tok / N, n = tok - b * N;
  const bool live = 2 * lane < D, pad = 2 * lane >= D && 2 * lane < Dp;
  float c0 = 0.f, s0 = 0.f, c1 = 0.f, s1 = 0.f, wq0 = 0.f, wq1 = 0.f, wk0 = 0.f, wk1 = 0.f;
  if (live) {
    c0 = cosb[(size_t)n * D + 2 * lane]; c1 = cosb[(size_t)n * D + 2 * lane + 1];
    s0 = sinb[(size_t)n * D + 2 * lane]; s1 = sinb[(size_t)n * D + 2 * lane + 1];
    wq0 = qw[2 * lane]; wq1 = qw[2 * lane + 1]; wk0 = kw[2 * lane]; wk1 = kw[2 * lane + 1];
  }
  const bf16* base = qkv + (size_t)tok * 3 * H * D;
  for (int h = 0; h < H; h++) {
    float q0 = 0.f, q1 = 0.f, k0 = 0.f, k1 = 0.f;
    bf16x2 vv = {(bf16)0.f, (bf16)0.f};
    if (live) {
      const bf16x2 a = *reinterpret_cast<const bf16x2*>(base + (size_t)h * D + 2 * lane);
      const bf16x2 c = *reinterpret_cast<const bf16x2*>(base + (size_t)(H + h) * D + 2 * lane);
      vv = *reinterpret_cast<const bf16x2*>(base + (size_t)(2 * H + h) * D + 2 * lane);
      q0 = (float)a[0]; q1 = (float)a[1]; k0 = (float)c[0]; k1 = (float)c[1];
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

static inline int grid_for(size_t n, int block = 256, int cap = 4096) {
  size_t g = (n + block - 1) / block;
  return (int)(g > (size_t)cap ? cap : (g < 1 ? 1 : g));
}

}  // namespace dmvae_dit
using namespace dmvae_dit;

extern "C" int dmvae_rmsnorm_modulate_bf16(const void* x, const void* w, const void* mod, void* y, int rows, int rows_per_sample, int c, int mod_stride,
                                           int shift_off, int scale_off, float eps, hipStream_t stream) {
  DMVAE_CHECK_ARG(x && w && mod && y && rows > 0 && rows_per_sample > 0, "rmsnorm_modulate_bf16: bad argument");
  DMVAE_CHECK_ARG(c % 4 == 0 && c >= 4 && c <= MAX_SWEEPS * 256, "rmsnorm_modulate_bf16: width must be a multiple of 4 up to 2048 (got %d)", c);
  DMVAE_CHECK_ARG(scale_off >= 0 && scale_off % 4 == 0 && (shift_off < 0 || shift_off % 4 == 0) && mod_stride % 4 == 0 && scale_off + c <= mod_stride,
                  "rmsnorm_modulate_bf16: modulation offsets must be multiples of 4 inside the row");
  hipLaunchKernelGGL(rmsnorm_modulate_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, (const float*)x, (const float*)w, (const bf16*)mod, (bf16*)y,
                     rows, rows_per_sample, c, mod_stride, shift_off, scale_off, eps);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

extern "C" int dmvae_qknorm_rope_bf16(const void* qkv, const void* q_weight, const void* k_weight, const void* cos_table, const void* sin_table,
                                      void* q_out, void* k_out, void* v_out, int batch, int seq, int heads, int head_dim, int head_dim_padded,
                                      float eps, hipStream_t stream) {
  DMVAE_CHECK_ARG(qkv && q_weight && k_weight && cos_table && sin_table && q_out && k_out && v_out && batch > 0 && seq > 0 && heads > 0,
                  "qknorm_rope_bf16: bad argument");
  DMVAE_CHECK_ARG(head_dim % 2 == 0 && head_dim >= 2 && head_dim_padded >= head_dim && head_dim_padded % 2 == 0 && head_dim_padded <= 128,
                  "qknorm_rope_bf16: head dim must be even, padded head dim <= 128 (got %d, %d)", head_dim, head_dim_padded);
  const int tokens = batch * seq;
  hipLaunchKernelGGL(qknorm_rope_kernel, dim3((tokens + 3) / 4), dim3(256), 0, stream, (const bf16*)qkv, (const float*)q_weight, (const float*)k_weight,
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
