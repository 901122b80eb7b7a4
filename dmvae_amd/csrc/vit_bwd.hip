// Backward-side elementwise / reduction kernels of the ViT encoder block (timm VisionTransformer as reached through
// models/vae.py:47-53; block algebra as in the reference's vendored models/dinov2.py + dino_layers/{block,attention,mlp,layer_scale}.py)
// for the stages where the encoder trains (train_dmd.py:349,519: vae-turn with requires_grad on the encoder).  The residual stream is
// f32 (LayerScale's f32 gamma promotes it under autocast), everything a Linear consumes or produces is bf16:
//
//   t  -> LN1 -> qkv -> attention -> proj -> t += ls1 * .  -> LN2 -> fc1 -> GELU -> fc2 -> t += ls2 * .
//
//   layernorm_bwd   : dt += rstd * (g - mean(g) - x_hat * mean(g * x_hat)),  g = dy * gamma;  dgamma += dy * x_hat, dbeta += dy
//   layerscale_bwd  : dy = gamma * dt (bf16);  dgamma += sum_rows dt * y
//   gelu fwd / bwd  : exact erf form (nn.GELU() default), f32 inside, bf16 in / out
//
// All HBM-bound: one pass over the operands, 16-B accesses, per-block partial sums + a fixed-order second stage (deterministic).
// The four Linear GEMMs per block (forward, input-gradient and weight-gradient) are plain library GEMMs (hipBLASLt through
// torch.matmul); attention backward is composed from the batched GEMM / softmax kernels of the decoder's AttnBlock.
#include "common.h"
#include "dmvae_hip.h"

namespace dmvae_vit_bwd {

constexpr int LN_MAX_BLOCKS = 256;

// One wave per row, rows dealt round-robin to the grid's waves; C = SWEEPS * 256.
// part: [gridDim.x][2][C] -- per-block sums of dy * x_hat (dgamma) and dy (dbeta).
template <int SWEEPS>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const bf16* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
                                                            float* __restrict__ dx_io, float* __restrict__ part, int rows, float eps) {
  constexpr int C = SWEEPS * 256;
  __shared__ float red[4][2][C];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  f32x4 gm[SWEEPS], ag[SWEEPS], ab[SWEEPS];
#pragma unroll
  for (int k = 0; k < SWEEPS; k++) {
    gm[k] = *reinterpret_cast<const f32x4*>(gamma + k * 256 + lane * 4);
    ag[k] = f32x4{0, 0, 0, 0}; ab[k] = f32x4{0, 0, 0, 0};
  }
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
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
      for (int e = 0; e < 4; e++) { v[k][e] -= mean; ss += v[k][e] * v[k][e]; }
    const float rstd = rsqrtf(wave_sum(ss) * (1.f / C) + eps);
    f32x4 g[SWEEPS];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < SWEEPS; k++) {
      const bf16x4 d = *reinterpret_cast<const bf16x4*>(dy + (size_t)row * C + k * 256 + lane * 4);
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const float xh = v[k][e] * rstd, dv = (float)d[e];
        v[k][e] = xh;
        g[k][e] = dv * gm[k][e];
        s1 += g[k][e]; s2 += g[k][e] * xh;
        ag[k][e] += dv * xh; ab[k][e] += dv;
      }
    }
    const float m1 = wave_sum(s1) * (1.f / C), m2 = wave_sum(s2) * (1.f / C);
    float* dr = dx_io + (size_t)row * C;
#pragma unroll
    for (int k = 0; k < SWEEPS; k++) {
      f32x4 o = *reinterpret_cast<const f32x4*>(dr + k * 256 + lane * 4);
#pragma unroll
      for (int e = 0; e < 4; e++) o[e] += rstd * (g[k][e] - m1 - v[k][e] * m2);
      *reinterpret_cast<f32x4*>(dr + k * 256 + lane * 4) = o;
    }
  }
#pragma unroll
  for (int k = 0; k < SWEEPS; k++) {
    *reinterpret_cast<f32x4*>(&red[wave][0][k * 256 + lane * 4]) = ag[k];
    *reinterpret_cast<f32x4*>(&red[wave][1][k * 256 + lane * 4]) = ab[k];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    const int w = i / C, c = i - w * C;
    part[((size_t)blockIdx.x * 2 + w) * C + c] = (red[0][w][c] + red[1][w][c]) + (red[2][w][c] + red[3][w][c]);
  }
}

// out[j][c] (+)= sum_b part[b][j][c], j < nj.  Block = 64 consecutive columns x 4 interleaved partial groups (coalesced 256-B rows,
// nblk/4 serial loads per thread), combined in a fixed order: deterministic.
__global__ __launch_bounds__(256) void colsum_parts_kernel(const float* __restrict__ part, float* __restrict__ o0, float* __restrict__ o1, int nblk,
                                                           int nj, int C, int accumulate) {
  __shared__ float red[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
  const int total = nj * C;
  float a = 0.f;
  if (col < total) {
    const int j = col / C, c = col - j * C;
#pragma unroll 8
    for (int b = grp; b < nblk; b += 4) a += part[((size_t)b * nj + j) * C + c];      // eight independent loads in flight (one at a time: 20 us for 64 steps); same order of additions
  }
  red[grp][threadIdx.x & 63] = a;
  __syncthreads();
  if (grp == 0 && col < total) {
    const int j = col / C, c = col - j * C;
    const float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    float* o = j == 0 ? o0 : o1;
    o[c] = (accumulate ? o[c] : 0.f) + v;
  }
}

// dy = gamma * dt (bf16); part[blk][c] = sum over the block's rows of dt * y.  One thread owns 8 channels; blockDim 256 = (C/8 lanes)
// x (2048/C rows) when C <= 2048.
__global__ __launch_bounds__(256) void layerscale_bwd_kernel(const float* __restrict__ dt, const bf16* __restrict__ y, const float* __restrict__ gamma,
                                                             bf16* __restrict__ dy, float* __restrict__ part, int rows, int C) {
  extern __shared__ float red[];  // [rows_per_block][C]
  const int c8 = C / 8, lanes = c8, rpb = 256 / lanes;
  const int lc = threadIdx.x % lanes, lr = threadIdx.x / lanes;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (lr < rpb) {
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + lc * 8), g1 = *reinterpret_cast<const f32x4*>(gamma + lc * 8 + 4);
    for (int row = blockIdx.x * rpb + lr; row < rows; row += gridDim.x * rpb) {
      const size_t off = (size_t)row * C + lc * 8;
      const f32x4 a = *reinterpret_cast<const f32x4*>(dt + off), b = *reinterpret_cast<const f32x4*>(dt + off + 4);
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(y + off);
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        o[e] = (bf16)(g0[e] * a[e]); o[4 + e] = (bf16)(g1[e] * b[e]);
        acc[e] += a[e] * (float)v[e]; acc[4 + e] += b[e] * (float)v[4 + e];
      }
      *reinterpret_cast<bf16x8*>(dy + off) = o;
    }
#pragma unroll
    for (int e = 0; e < 8; e++) red[lr * C + lc * 8 + e] = acc[e];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = 0.f;
    for (int r = 0; r < rpb; r++) a += red[r * C + c];
    part[(size_t)blockIdx.x * C + c] = a;
  }
}

__device__ __forceinline__ float gelu_f(float x) { return dmvae_gelu_f(x); }   // common.h: erf from A&S 7.1.26, the same bits as the GEMM epilogue's
__device__ __forceinline__ float gelu_grad_f(float x) {
  return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

__global__ void gelu_fwd_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, size_t n8) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    const bf16x8 v = reinterpret_cast<const bf16x8*>(x)[i];
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; e++) o[e] = (bf16)gelu_f((float)v[e]);
    reinterpret_cast<bf16x8*>(y)[i] = o;
  }
}
__global__ void gelu_bwd_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x, bf16* __restrict__ dx, size_t n8) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    const bf16x8 v = reinterpret_cast<const bf16x8*>(x)[i], d = reinterpret_cast<const bf16x8*>(dy)[i];
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; e++) o[e] = (bf16)((float)d[e] * gelu_grad_f((float)v[e]));
    reinterpret_cast<bf16x8*>(dx)[i] = o;
  }
}

static inline int grid_for(size_t n, int block = 256, int cap = 4096) {
  size_t g = (n + block - 1) / block;
  return (int)(g > (size_t)cap ? cap : (g < 1 ? 1 : g));
}

}  // namespace dmvae_vit_bwd
using namespace dmvae_vit_bwd;

extern "C" size_t dmvae_vit_bwd_workspace(int c) { return (size_t)LN_MAX_BLOCKS * 2 * (size_t)c * sizeof(float); }

extern "C" int dmvae_layernorm_bwd_f32(const void* dy, const void* x, const void* gamma, void* dx_io, void* dgamma, void* dbeta, void* workspace,
                                       size_t workspace_bytes, int rows, int c, float eps, int accumulate, hipStream_t stream) {
  DMVAE_CHECK_ARG(dy && x && gamma && dx_io && workspace && rows > 0, "layernorm_bwd_f32: bad argument");
  DMVAE_CHECK_ARG(c == 256 || c == 512 || c == 768 || c == 1024 || c == 1280 || c == 1536,
                  "layernorm_bwd_f32: width must be a multiple of 256 up to 1536 (got %d)", c);
  DMVAE_CHECK_ARG(workspace_bytes >= dmvae_vit_bwd_workspace(c), "layernorm_bwd_f32: workspace too small");
  int nblk = (rows + 3) / 4; if (nblk > LN_MAX_BLOCKS) nblk = LN_MAX_BLOCKS;
  const dim3 grid(nblk), block(256);
#define DMVAE_LNB(S) hipLaunchKernelGGL(layernorm_bwd_kernel<S>, grid, block, 0, stream, (const bf16*)dy, (const float*)x, (const float*)gamma, \
                                        (float*)dx_io, (float*)workspace, rows, eps)
  switch (c / 256) {
    case 1: DMVAE_LNB(1); break;
    case 2: DMVAE_LNB(2); break;
    case 3: DMVAE_LNB(3); break;
    case 4: DMVAE_LNB(4); break;
    case 5: DMVAE_LNB(5); break;
    default: DMVAE_LNB(6); break;
  }
#undef DMVAE_LNB
  DMVAE_CHECK_LAUNCH();
  if (dgamma && dbeta) {
    hipLaunchKernelGGL(colsum_parts_kernel, dim3((2 * c + 63) / 64), dim3(256), 0, stream, (const float*)workspace, (float*)dgamma, (float*)dbeta,
                       nblk, 2, c, accumulate);
    DMVAE_CHECK_LAUNCH();
  }
  return 0;
}

extern "C" int dmvae_layerscale_bwd(const void* dt, const void* y, const void* gamma, void* dy, void* dgamma, void* workspace,
                                    size_t workspace_bytes, int rows, int c, int accumulate, hipStream_t stream) {
  DMVAE_CHECK_ARG(dt && y && gamma && dy && dgamma && workspace && rows > 0, "layerscale_bwd: bad argument");
  DMVAE_CHECK_ARG(c % 8 == 0 && c >= 8 && c <= 2048, "layerscale_bwd: width must be a multiple of 8 up to 2048 (got %d)", c);   // c / 8 lanes per row, 256 / (c / 8) rows per block (ViT-B: 96 lanes, 2 rows, 64 idle threads)
  DMVAE_CHECK_ARG(workspace_bytes >= dmvae_vit_bwd_workspace(c), "layerscale_bwd: workspace too small");
  const int rpb = 256 / (c / 8);
  int nblk = (rows + rpb * 8 - 1) / (rpb * 8); if (nblk > LN_MAX_BLOCKS) nblk = LN_MAX_BLOCKS; if (nblk < 1) nblk = 1;
  hipLaunchKernelGGL(layerscale_bwd_kernel, dim3(nblk), dim3(256), (size_t)rpb * c * sizeof(float), stream, (const float*)dt, (const bf16*)y,
                     (const float*)gamma, (bf16*)dy, (float*)workspace, rows, c);
  DMVAE_CHECK_LAUNCH();
  hipLaunchKernelGGL(colsum_parts_kernel, dim3((c + 63) / 64), dim3(256), 0, stream, (const float*)workspace, (float*)dgamma, (float*)dgamma, nblk, 1, c,
                     accumulate);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

extern "C" int dmvae_gelu_fwd(const void* x, void* y, size_t n, hipStream_t stream) {
  DMVAE_CHECK_ARG(x && y && n % 8 == 0, "gelu_fwd: element count must be a multiple of 8");
  if (n == 0) return 0;
  hipLaunchKernelGGL(gelu_fwd_kernel, dim3(grid_for(n / 8)), dim3(256), 0, stream, (const bf16*)x, (bf16*)y, n / 8);
  DMVAE_CHECK_LAUNCH();
  return 0;
}
extern "C" int dmvae_gelu_bwd(const void* dy, const void* x, void* dx, size_t n, hipStream_t stream) {
  DMVAE_CHECK_ARG(dy && x && dx && n % 8 == 0, "gelu_bwd: element count must be a multiple of 8");
  if (n == 0) return 0;
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3(grid_for(n / 8)), dim3(256), 0, stream, (const bf16*)dy, (const bf16*)x, (bf16*)dx, n / 8);
  DMVAE_CHECK_LAUNCH();
  return 0;
}
