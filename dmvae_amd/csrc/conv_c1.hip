// 4x4 stride-1 padding-1 convolution to ONE output channel: the PatchGAN's logits layer (models/patchgan.py:146 of the reference, nn.Conv2d(512, 1, 4, 1, 1) on the
// 31 x 31 x 512 map -> 30 x 30 logits), forward, input gradient and weight / bias gradient, NHWC bf16 activations, f32 logits.
//
// On the matrix-core kernels this layer is a GEMM with one useful output row (the forward pads it to four, the input gradient pads dY's single channel to a
// 32-channel reduction, the weight gradient to a 32-row product): 174 / 465 / 135 us per call at B = 64 against 63 MB of activations -- 2.1 ms of the adversarial
// step for 1 GFLOP per pass.  By its bytes the layer is one read (or one write) of the activation map, and by its arithmetic 8 192 multiply-adds per logit: VALU
// work.  Here a WAVE owns one image row and a LANE eight channels of every pixel (64 lanes x 8 = 512 channels = one 1-KiB pixel per wave load):
//   forward   out[y][xo] = sum over ky, ix, kx of <x[y + ky - 1][ix][8 ch], w[ky][kx][8 ch]> into 30 per-lane accumulators (xo = ix - kx + 1), then a butterfly
//             over the 64 lanes per logit;
//   dgrad     dx[iy][ix][8 ch] = sum over ky, kx of dy[iy - ky + 1][ix - kx + 1] * w[ky][kx][8 ch]: the dy row lives one value per lane and is broadcast by
//             v_readlane with a compile-time lane index (the pixel loop is fully unrolled);
//   wgrad     dw[ky][kx][8 ch] += dy[iy - ky + 1][ix - kx + 1] * x[iy][ix][8 ch] into 128 accumulators per lane over the rows a wave is dealt (fixed deal), the
//             waves' partials summed in a fixed order by a second kernel that also lays the result out as the parameter [1][C][4][4]; db = sum dy there.
// The weights are read from the f32 parameter and rounded to bf16 in registers (what the autocast conv multiplies by); products and sums in f32.
// Shapes: W = 31 (compile-time: the unrolled pixel loop), C a multiple of 512, any N, H; everything else stays on the general kernels.  Deterministic.
#include "common.h"
#include "dmvae_hip.h"

namespace dmvae_c1 {

constexpr int W_IN = 31, W_OUT = 30;

__device__ __forceinline__ void cvt8(const uint4& u, float (&f)[8]) {
  const bf16x8 v = *reinterpret_cast<const bf16x8*>(&u);
#pragma unroll
  for (int i = 0; i < 8; i++) f[i] = (float)v[i];
}
// the lane's eight channels of tap (ky, kx) from the parameter [1][C][4][4] f32, rounded to bf16 as the autocast conv's operand
__device__ __forceinline__ void load_w(const float* __restrict__ w, int c0, float (&wr)[16][8]) {
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const f32x4* p = reinterpret_cast<const f32x4*>(w + (size_t)(c0 + i) * 16);
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const f32x4 t = p[q];
#pragma unroll
      for (int e = 0; e < 4; e++) wr[q * 4 + e][i] = (float)(bf16)t[e];
    }
  }
}

// grid: ceil(N * Ho / 4) blocks of 4 waves; wave -> output row r = (n, y)
__global__ __launch_bounds__(256) void fwd_kernel(const bf16* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ out,
                                                  int N, int H, int C) {
  const int Ho = H - 1, lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= N * Ho) return;
  const int n = r / Ho, y = r - n * Ho;
  float acc[W_OUT];
#pragma unroll
  for (int i = 0; i < W_OUT; i++) acc[i] = 0.f;
  for (int c0 = lane * 8; c0 < C; c0 += 512) {
#pragma unroll 1
    for (int ky = 0; ky < 4; ky++) {      // NOT unrolled: the body is 1 700 instructions; four copies of it (50 KB of straight-line code run once per wave) starve on instruction fetch
      const int iy = y + ky - 1;
      if (iy < 0 || iy >= H) continue;      // wave-uniform
      // the row's 31 pixels requested together, ahead of the arithmetic (left to the compiler -- 128 weight + 30 sum registers live -- each pixel's load sat
      // directly in front of its 32 multiply-adds: 124 serial trips to memory per logit row); only this kernel row's 4 x 8 weights are held
      const bf16* row = x + ((size_t)(n * H + iy) * W_IN) * C + c0;
      uint4 px[W_IN];
#pragma unroll
      for (int ix = 0; ix < W_IN; ix++) px[ix] = *reinterpret_cast<const uint4*>(row + (size_t)ix * C);
      float wr[4][8];
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(w + (size_t)(c0 + i) * 16 + ky * 4);
#pragma unroll
        for (int e = 0; e < 4; e++) wr[e][i] = (float)(bf16)t[e];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ix = 0; ix < W_IN; ix++) {
        float v[8];
        cvt8(px[ix], v);
#pragma unroll
        for (int kx = 0; kx < 4; kx++) {
          const int xo = ix - kx + 1;
          if (xo < 0 || xo >= W_OUT) continue;      // compile-time
#pragma unroll
          for (int i = 0; i < 8; i++) acc[xo] = fmaf(v[i], wr[kx][i], acc[xo]);
        }
      }
    }
  }
  const float b = bias ? bias[0] : 0.f;
  // 30 sums over the 64 lanes in 32 exchanges instead of 180: every step a lane hands the half of its values its partner keeps and adds the partner's half of the
  // ones it keeps (32 -> 16 -> ... -> 1 value per lane over lane bits 5 .. 1), then one exchange over bit 0: lanes 2 k and 2 k + 1 hold logit k
  float v[32];
#pragma unroll
  for (int i = 0; i < 32; i++) v[i] = i < W_OUT ? acc[i] : 0.f;
#pragma unroll
  for (int o = 32, cnt = 32; o >= 2; o >>= 1, cnt >>= 1) {
    const bool upper = (lane & o) != 0;
#pragma unroll
    for (int i = 0; i < cnt / 2; i++) {
      const float send = upper ? v[i] : v[i + cnt / 2], keep = upper ? v[i + cnt / 2] : v[i];
      v[i] = keep + __shfl_xor(send, o, 64);
    }
  }
  const float s = v[0] + __shfl_xor(v[0], 1, 64);
  if ((lane & 1) == 0 && (lane >> 1) < W_OUT) out[(size_t)r * W_OUT + (lane >> 1)] = s + b;
}

// grid: ceil(N * H / 4) blocks; wave -> input row (n, iy); dy f32 [N][Ho][30]
__global__ __launch_bounds__(256) void dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w, bf16* __restrict__ dx, int N, int H, int C) {
  const int Ho = H - 1, lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= N * H) return;
  const int n = r / H, iy = r - n * H;
  float dyr[4];      // lane xo holds dy[iy - ky + 1][xo] (zero outside)
#pragma unroll
  for (int ky = 0; ky < 4; ky++) {
    const int yo = iy - ky + 1;
    dyr[ky] = (yo >= 0 && yo < Ho && lane < W_OUT) ? dy[((size_t)n * Ho + yo) * W_OUT + lane] : 0.f;
  }
  for (int c0 = lane * 8; c0 < C; c0 += 512) {
    float wr[16][8];
    load_w(w, c0, wr);
    bf16* row = dx + ((size_t)(n * H + iy) * W_IN) * C + c0;
    // Pixels unrolled (the broadcast's lane index is then an immediate): 3 800 instructions, 28 KB -- measured 23.7 us against 30.7 us as a loop with the index in a
    // scalar register.  (The forward kernel is the opposite case: with its four kernel rows unrolled as well it was 50 KB of straight-line code run once per wave
    // and spent its time fetching it, 85 us; with the rows as a loop 37 us.)  xo = ix - kx + 1 runs over -2 .. 31: lanes 30 .. 63 of dyr hold zeros, so the
    // out-of-range taps read a zero through (xo & 63) and need no branch.
#pragma unroll
    for (int ix = 0; ix < W_IN; ix++) {
      float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ky = 0; ky < 4; ky++)
#pragma unroll
        for (int kx = 0; kx < 4; kx++) {
          const float g = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dyr[ky]), (ix - kx + 1) & 63));
#pragma unroll
          for (int i = 0; i < 8; i++) a[i] = fmaf(g, wr[ky * 4 + kx][i], a[i]);
        }
      bf16x8 o;
#pragma unroll
      for (int i = 0; i < 8; i++) o[i] = (bf16)a[i];
      *reinterpret_cast<bf16x8*>(row + (size_t)ix * C) = o;
    }
  }
}

// grid: WG_BLOCKS blocks of 4 waves; wave g of G = 4 * gridDim.x takes the input rows g, g + G, ... (fixed deal); part: [WG_BLOCKS][16][C] f32 (one per block)
constexpr int WG_BLOCKS = 256;
__global__ __launch_bounds__(256) void wgrad_kernel(const bf16* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part, float* __restrict__ bpart, int N, int H,
                                                    int C) {
  const int Ho = H - 1, lane = threadIdx.x & 63;
  const int g = blockIdx.x * 4 + (threadIdx.x >> 6), G = gridDim.x * 4;
  float bsum = 0.f;      // the wave's share of db = sum dy: lane xo adds dy[iy][xo] of the rows it is dealt (first channel chunk only)
  for (int c0 = lane * 8; c0 < C; c0 += 512) {
    float acc[16][8];
#pragma unroll
    for (int t = 0; t < 16; t++)
#pragma unroll
      for (int i = 0; i < 8; i++) acc[t][i] = 0.f;
    for (int r = g; r < N * H; r += G) {
      const int n = r / H, iy = r - n * H;
      float dyr[4];
#pragma unroll
      for (int ky = 0; ky < 4; ky++) {
        const int yo = iy - ky + 1;
        dyr[ky] = (yo >= 0 && yo < Ho && lane < W_OUT) ? dy[((size_t)n * Ho + yo) * W_OUT + lane] : 0.f;
      }
      if (c0 < 512) bsum += dyr[1];      // ky = 1: yo = iy
      const bf16* row = x + ((size_t)(n * H + iy) * W_IN) * C + c0;
      // pixels in groups of four, a loop (see dgrad_kernel), the next group's loads in flight under the current group's arithmetic
      uint4 cur[4], nxt[4];
#pragma unroll
      for (int k = 0; k < 4; k++) cur[k] = *reinterpret_cast<const uint4*>(row + (size_t)k * C);
#pragma unroll 1
      for (int ix0 = 0; ix0 < W_IN; ix0 += 4) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
          nxt[k] = uint4{0, 0, 0, 0};
          if (ix0 + 4 + k < W_IN) nxt[k] = *reinterpret_cast<const uint4*>(row + (size_t)(ix0 + 4 + k) * C);      // scalar condition
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
          float v[8];
          cvt8(cur[k], v);       // pixels past the row's end are zeros: they add nothing
#pragma unroll
          for (int ky = 0; ky < 4; ky++)
#pragma unroll
            for (int kx = 0; kx < 4; kx++) {
              const float gq = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dyr[ky]), (ix0 + k - kx + 1) & 63));
#pragma unroll
              for (int i = 0; i < 8; i++) acc[ky * 4 + kx][i] = fmaf(gq, v[i], acc[ky * 4 + kx][i]);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) cur[k] = nxt[k];
      }
    }
    // the block's four waves: summed in wave order through LDS (eight taps at a time: 48 KiB), one partial per block
    __shared__ float red[3][8][512];
    const int wv = threadIdx.x >> 6;
#pragma unroll
    for (int th = 0; th < 2; th++) {
      __syncthreads();
      if (wv > 0) {
#pragma unroll
        for (int t = 0; t < 8; t++)
#pragma unroll
          for (int i = 0; i < 8; i++) red[wv - 1][t][i * 64 + lane] = acc[th * 8 + t][i];
      }
      __syncthreads();
      if (wv == 0) {
#pragma unroll
        for (int t = 0; t < 8; t++) {
          float o[8];
#pragma unroll
          for (int i = 0; i < 8; i++) o[i] = ((acc[th * 8 + t][i] + red[0][t][i * 64 + lane]) + red[1][t][i * 64 + lane]) + red[2][t][i * 64 + lane];
          float* p = part + ((size_t)blockIdx.x * 16 + th * 8 + t) * C + c0;
          *reinterpret_cast<f32x4*>(p) = f32x4{o[0], o[1], o[2], o[3]};
          *reinterpret_cast<f32x4*>(p + 4) = f32x4{o[4], o[5], o[6], o[7]};
        }
      }
    }
  }
  if (bpart) {
    __shared__ float bred[4];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) bsum += __shfl_xor(bsum, o, 64);
    if (lane == 0) bred[threadIdx.x >> 6] = bsum;
    __syncthreads();
    if (threadIdx.x == 0) bpart[blockIdx.x] = ((bred[0] + bred[1]) + bred[2]) + bred[3];
  }
}
// dw[0][c][ky][kx] = sum over the P block partials: sixteen lanes per element take partials j, j + 16, ... in order and are summed by a fixed butterfly; block 0
// also sums the blocks' bias partials into db (a serial sum of dy's 57 600 values in one block was 60 us: the whole kernel's duration).   grid: ceil(16 * C * 16 / 256)
__global__ __launch_bounds__(256) void wgrad_final_kernel(const float* __restrict__ part, const float* __restrict__ bpart, float* __restrict__ dw, float* __restrict__ db,
                                                          int P, int C) {
  const int gi = blockIdx.x * 256 + threadIdx.x, i = gi >> 4, j = gi & 15;      // i = t * C + c
  float s = 0.f;
  if (i < 16 * C)
    for (int p = j; p < P; p += 16) s += part[(size_t)p * 16 * C + i];
#pragma unroll
  for (int o = 8; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
  if (i < 16 * C && j == 0) {
    const int t = i / C, c = i - t * C;
    dw[(size_t)c * 16 + t] = s;
  }
  if (db && blockIdx.x == 0) {      // db = the P block partials in a fixed tree (P <= 256)
    __shared__ float red[256];
    red[threadIdx.x] = (int)threadIdx.x < P ? bpart[threadIdx.x] : 0.f;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
      if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) db[0] = red[0];
  }
}

}  // namespace dmvae_c1

extern "C" int dmvae_conv_k4c1_supported(int n, int h, int w, int c) { return (n > 0 && h >= 2 && w == dmvae_c1::W_IN && c >= 512 && c % 512 == 0) ? 1 : 0; }
extern "C" size_t dmvae_conv_k4c1_wgrad_workspace(int c) { return (size_t)dmvae_c1::WG_BLOCKS * (16 * (size_t)c + 1) * sizeof(float); }

extern "C" int dmvae_conv_k4c1_fwd(const void* x, const void* w, const void* bias, void* out, int n, int h, int wdt, int c, hipStream_t stream) {
  DMVAE_CHECK_ARG(x && w && out && dmvae_conv_k4c1_supported(n, h, wdt, c), "conv_k4c1_fwd: x [%d][%d][%d][%d] (W = 31, C a multiple of 512)", n, h, wdt, c);
  hipLaunchKernelGGL(dmvae_c1::fwd_kernel, dim3((n * (h - 1) + 3) / 4), dim3(256), 0, stream, (const bf16*)x, (const float*)w, (const float*)bias, (float*)out, n, h, c);
  DMVAE_CHECK_LAUNCH();
  return 0;
}
extern "C" int dmvae_conv_k4c1_dgrad(const void* dy, const void* w, void* dx, int n, int h, int wdt, int c, hipStream_t stream) {
  DMVAE_CHECK_ARG(dy && w && dx && dmvae_conv_k4c1_supported(n, h, wdt, c), "conv_k4c1_dgrad: x [%d][%d][%d][%d] (W = 31, C a multiple of 512)", n, h, wdt, c);
  hipLaunchKernelGGL(dmvae_c1::dgrad_kernel, dim3((n * h + 3) / 4), dim3(256), 0, stream, (const float*)dy, (const float*)w, (bf16*)dx, n, h, c);
  DMVAE_CHECK_LAUNCH();
  return 0;
}
extern "C" int dmvae_conv_k4c1_wgrad(const void* x, const void* dy, void* dw, void* db, void* workspace, size_t workspace_bytes, int n, int h, int wdt, int c,
                                     hipStream_t stream) {
  DMVAE_CHECK_ARG(x && dy && dw && workspace && dmvae_conv_k4c1_supported(n, h, wdt, c), "conv_k4c1_wgrad: x [%d][%d][%d][%d] (W = 31, C a multiple of 512)", n, h, wdt, c);
  DMVAE_CHECK_ARG(workspace_bytes >= dmvae_conv_k4c1_wgrad_workspace(c), "conv_k4c1_wgrad: workspace too small");
  using namespace dmvae_c1;
  float* bpart = (float*)workspace + (size_t)WG_BLOCKS * 16 * c;
  hipLaunchKernelGGL(wgrad_kernel, dim3(WG_BLOCKS), dim3(256), 0, stream, (const bf16*)x, (const float*)dy, (float*)workspace, db ? bpart : nullptr, n, h, c);
  DMVAE_CHECK_LAUNCH();
  hipLaunchKernelGGL(wgrad_final_kernel, dim3((16 * c * 16 + 255) / 256), dim3(256), 0, stream, (const float*)workspace, (const float*)bpart, (float*)dw, (float*)db, WG_BLOCKS, c);
  DMVAE_CHECK_LAUNCH();
  return 0;
}
