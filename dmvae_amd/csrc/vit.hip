// Elementwise kernels of the frozen ViT encoder forward (models/vae.py:52-53 -> timm VisionTransformer blocks; block algebra as
// in the reference's vendored models/dino_layers/block.py:89-115): the f32 residual stream stays in HBM, everything that
// feeds a GEMM is bf16.  Both are HBM-bound row kernels (16-B accesses, one wave per row, f32 statistics).
//   layernorm_f32_bf16 : y = bf16( (x - mean) * rstd * gamma + beta )      replaces nn.LayerNorm (f32 under autocast) + the bf16 cast
//   scale_residual_f32 : x += gamma * float(y)                              replaces LayerScale (dino_layers/layer_scale.py:15-26) + the
//                                                                          residual add (f32 under autocast type promotion)
#include "common.h"
#include "dmvae_hip.h"

namespace dmvae_vit {

// one wave per row; C % 256 == 0 (4 floats per lane per sweep), C <= 4096
template <int SWEEPS>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, bf16* __restrict__ y, int rows, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  constexpr int C = SWEEPS * 256;
  const float* xr = x + (size_t)row * C;
  f32x4 v[SWEEPS];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < SWEEPS; k++) {
    v[k] = *reinterpret_cast<const f32x4*>(xr + k * 256 + lane * 4);
    s += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
  }
  const float mean = wave_sum(s) * (1.f / C);
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < SWEEPS; k++)
#pragma unroll
    for (int e = 0; e < 4; e++) { const float d = v[k][e] - mean; ss += d * d; }
  const float rstd = rsqrtf(wave_sum(ss) * (1.f / C) + eps);
  bf16* yr = y + (size_t)row * C;
#pragma unroll
  for (int k = 0; k < SWEEPS; k++) {
    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + k * 256 + lane * 4);
    const f32x4 b = *reinterpret_cast<const f32x4*>(beta + k * 256 + lane * 4);
    bf16x4 o;
#pragma unroll
    for (int e = 0; e < 4; e++) o[e] = (bf16)((v[k][e] - mean) * rstd * g[e] + b[e]);
    *reinterpret_cast<bf16x4*>(yr + k * 256 + lane * 4) = o;
  }
}

__global__ __launch_bounds__(256) void scale_residual_kernel(float* __restrict__ x, const bf16* __restrict__ y, const float* __restrict__ gamma,
                                                             size_t n8, int c8) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % c8) * 8;
    const bf16x8 v = reinterpret_cast<const bf16x8*>(y)[i];
    f32x4 a = reinterpret_cast<const f32x4*>(x)[2 * i], b = reinterpret_cast<const f32x4*>(x)[2 * i + 1];
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + c), g1 = *reinterpret_cast<const f32x4*>(gamma + c + 4);
#pragma unroll
    for (int e = 0; e < 4; e++) { a[e] = fmaf(g0[e], (float)v[e], a[e]); b[e] = fmaf(g1[e], (float)v[4 + e], b[e]); }
    reinterpret_cast<f32x4*>(x)[2 * i] = a;
    reinterpret_cast<f32x4*>(x)[2 * i + 1] = b;
  }
}

// P[r][:] = softmax(scale * S[r][:]) for bf16 scores (the rounded QK^T of the encoder's attention), f32 inside, bf16 out; one wave
// per row, cols <= 512.  Replaces the scale multiply, the f32 up-cast, softmax and the bf16 down-cast (four ATen kernels).
__global__ __launch_bounds__(256) void softmax_bf16_kernel(const bf16* __restrict__ s, bf16* __restrict__ p, size_t rows, int cols, float scale) {
  const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const bf16* sr = s + row * cols;
  float v[8];
  float m = -INFINITY;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int c = lane + k * 64;
    v[k] = c < cols ? (float)sr[c] * scale : -INFINITY;
    m = fmaxf(m, v[k]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < 8; k++) { v[k] = __expf(v[k] - m); sum += v[k]; }
  const float inv = 1.f / wave_sum(sum);
  bf16* pr = p + row * cols;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int c = lane + k * 64;
    if (c < cols) pr[c] = (bf16)(v[k] * inv);
  }
}

}  // namespace dmvae_vit
using namespace dmvae_vit;

extern "C" int dmvae_layernorm_f32_bf16(const void* x, const void* gamma, const void* beta, void* y, int rows, int c, float eps,
                                        hipStream_t stream) {
  DMVAE_CHECK_ARG(x && gamma && beta && y && rows > 0, "layernorm_f32_bf16: bad argument");
  DMVAE_CHECK_ARG(c == 256 || c == 512 || c == 768 || c == 1024 || c == 1280 || c == 1536,
                  "layernorm_f32_bf16: width must be a multiple of 256 up to 1536 (got %d)", c);
  const dim3 grid((rows + 3) / 4), block(256);
  switch (c / 256) {
    case 1: hipLaunchKernelGGL(layernorm_kernel<1>, grid, block, 0, stream, (const float*)x, (const float*)gamma, (const float*)beta, (bf16*)y, rows, eps); break;
    case 2: hipLaunchKernelGGL(layernorm_kernel<2>, grid, block, 0, stream, (const float*)x, (const float*)gamma, (const float*)beta, (bf16*)y, rows, eps); break;
    case 3: hipLaunchKernelGGL(layernorm_kernel<3>, grid, block, 0, stream, (const float*)x, (const float*)gamma, (const float*)beta, (bf16*)y, rows, eps); break;
    case 4: hipLaunchKernelGGL(layernorm_kernel<4>, grid, block, 0, stream, (const float*)x, (const float*)gamma, (const float*)beta, (bf16*)y, rows, eps); break;
    case 5: hipLaunchKernelGGL(layernorm_kernel<5>, grid, block, 0, stream, (const float*)x, (const float*)gamma, (const float*)beta, (bf16*)y, rows, eps); break;
    default: hipLaunchKernelGGL(layernorm_kernel<6>, grid, block, 0, stream, (const float*)x, (const float*)gamma, (const float*)beta, (bf16*)y, rows, eps); break;
  }
  DMVAE_CHECK_LAUNCH();
  return 0;
}

extern "C" int dmvae_scale_residual_f32(void* x, const void* y, const void* gamma, size_t rows, int c, hipStream_t stream) {
  DMVAE_CHECK_ARG(x && y && gamma && c > 0 && c % 8 == 0, "scale_residual_f32: width must be a multiple of 8");
  if (rows == 0) return 0;
  const size_t n8 = rows * (size_t)(c / 8);
  size_t nb = (n8 + 255) / 256; if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(scale_residual_kernel, dim3((unsigned)nb), dim3(256), 0, stream, (float*)x, (const bf16*)y, (const float*)gamma, n8, c / 8);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

extern "C" int dmvae_softmax_rows_bf16(const void* s, void* p, size_t rows, int cols, float scale, hipStream_t stream) {
  DMVAE_CHECK_ARG(s && p && cols > 0 && cols <= 512, "softmax_rows_bf16: cols must be in 1..512 (got %d)", cols);
  if (rows == 0) return 0;
  hipLaunchKernelGGL(softmax_bf16_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, (const bf16*)s, (bf16*)p, rows, cols, scale);
  DMVAE_CHECK_LAUNCH();
  return 0;
}
