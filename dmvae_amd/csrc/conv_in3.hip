// 3x3 stride-1 convolution FROM THREE input channels (an image) to 64 / 128 output channels, bias and ReLU in the epilogue, NHWC bf16 result: the first layer of
// the LPIPS trunk (VGG16 conv1_1 behind the ScalingLayer, utils/lpips.py:81-104,116-135 of the reference) on both branches at once.
//
// The general kernel runs this layer with its three channels zero-padded to one 32-channel K step per tap (K = 288, 91 % of it zeros): 264 us for the conv at
// 64 x 256 x 256, 96 us for the pass that writes the padded NHWC copy (268 MB) and ~40 us of elementwise launches for the ScalingLayer and the concatenation in
// front of it.  By the bytes the layer is its OUTPUT (537 MB): the image is 25 MB.  Here
//   * pad_kernel turns the NCHW f32 images (one or two source tensors: no concatenation) into a zero-bordered 4-channel bf16 copy [N][H + 2][W + 2][4] and applies
//     the ScalingLayer on the way ((x - shift) / scale in f32, then the bf16 rounding the padded route had): 34 MB, cache-resident;
//   * conv_in3_kernel: a wave owns 16-pixel runs of an image row.  K = taps 0-7 x 4 channels is ONE 32-deep MFMA step, tap 8 a second one: lane (p, kg) supplies,
//     as the B operand, the 8-B pixels of taps 2 kg and 2 kg + 1 of pixel p (unconditional loads: the border is in the copy); the weights (A operand, 16 output
//     channels per fragment) are wave-constant registers converted from the f32 parameter at block start.  Both steps are v_mfma_f32_16x16x32_bf16 (an accumulate
//     chain through two different MFMA opcodes is not interlocked: tools/probes/probe_mfma_chain.hip).  D (lane <-> pixel p, channels 16 f + 4 kg + i) gets bias
//     and ReLU, is rounded to bf16 and turned through a per-wave LDS tile into 1-KB contiguous wave stores.
// Store-bound: 2 bytes per output element + 0.13 per pixel of input.
#include "common.h"
#include "dmvae_hip.h"

namespace dmvae_cin3 {

__global__ void pad_kernel(const float* __restrict__ x0, const float* __restrict__ x1, int n0, const float* __restrict__ shift, const float* __restrict__ scale,
                           bf16* __restrict__ out, int N, int H, int W) {
  const size_t total = (size_t)N * (H + 2) * (W + 2);
  float sh[3] = {0.f, 0.f, 0.f}, sc[3] = {1.f, 1.f, 1.f};
  if (shift && scale) {
#pragma unroll
    for (int c = 0; c < 3; c++) { sh[c] = shift[c]; sc[c] = scale[c]; }
  }
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int xx = (int)(i % (W + 2)) - 1;
    const size_t r = i / (W + 2);
    const int yy = (int)(r % (H + 2)) - 1, n = (int)(r / (H + 2));
    bf16x4 v = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
    if (xx >= 0 && xx < W && yy >= 0 && yy < H) {
      const float* s = (n < n0 ? x0 + (size_t)n * 3 * H * W : x1 + (size_t)(n - n0) * 3 * H * W) + (size_t)yy * W + xx;
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const float t = s[(size_t)c * H * W];
        v[c] = (bf16)(shift && scale ? (t - sh[c]) / sc[c] : t);      // the reference's ScalingLayer arithmetic, utils/lpips.py:103
      }
    }
    *reinterpret_cast<bf16x4*>(out + i * 4) = v;
  }
}

template <int NF, bool RELU>      // output channels / 16
__global__ __launch_bounds__(256) void conv_in3_kernel(const bf16* __restrict__ xp, const float* __restrict__ wt, const float* __restrict__ bias,
                                                       bf16* __restrict__ y, int N, int H, int W) {
  constexpr int COUT = 16 * NF, ROW = COUT * 2 + 16;            // bytes per pixel row of a wave's tile
  constexpr int CPP = 2 * NF, PPI = 64 / CPP, ROUNDS = 16 / PPI;  // 16-B chunks per pixel, pixels per wave store, stores per run
  __shared__ __attribute__((aligned(16))) char smem[4 * 16 * ROW + 4 * 448];      // per wave: the result tile and the input stage (3 x 18 pixels of 8 B)
  const int wv = threadIdx.x >> 6, l = threadIdx.x & 63, p = l & 15, kg = l >> 4;
  char* tile = smem + wv * 16 * ROW;
  // A operands: row p of fragment f is output channel 16 f + p; step 1: k = 8 kg + kk <-> tap 2 kg + (kk >> 2), input channel kk & 3; step 2: tap 8 in k = 0..2
  bf16x8 wa[NF], wb[NF];
  f32x4 bs[NF];      // the bias is the accumulator's initial value
#pragma unroll
  for (int f = 0; f < NF; f++) {
    const int ch = 16 * f + p;
#pragma unroll
    for (int kk = 0; kk < 8; kk++) {
      const int ci = kk & 3, tap = 2 * kg + (kk >> 2);
      const int cc = ci < 3 ? ci : 2;                     // every load unconditional, every mask applied to the value: no branch, no wait between them
      const float v1 = wt[((size_t)ch * 3 + cc) * 9 + tap], v2 = wt[((size_t)ch * 3 + cc) * 9 + 8];
      wa[f][kk] = (bf16)(ci < 3 ? v1 : 0.f);
      wb[f][kk] = (bf16)((ci < 3 && kg == 0 && kk < 4) ? v2 : 0.f);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const float v = (bias ? bias : wt)[16 * f + 4 * kg + i];
      bs[f][i] = bias ? v : 0.f;
    }
  }
  // A run's input is 3 rows x 18 pixels of the bordered copy (432 B): lanes 0..53 load one 8-B pixel each (ONE load instruction per run, three contiguous
  // 144-B pieces), park it in the wave's LDS stage, and every lane picks its taps from there: tap t = (ky, kx) = (t / 3, t % 3) of output pixel p is stage
  // pixel (ky, p + kx).  (First form: each lane loaded its three tap pixels from global memory -- three instructions of 8 B per lane over 4-8 cache lines each:
  // with the stores they kept the CU's memory pipeline busy for 140 us; arithmetic alone 70, stores alone 85.)
  const int Wp = W + 2, HW = H * W;
  const int sr = l / 18, scol = l - sr * 18;                    // lane -> stage pixel (lanes >= 54 idle)
  const int t0 = 2 * kg, t1 = 2 * kg + 1;
  char* stage = smem + 4 * 16 * ROW + wv * 448;
  const int so0 = ((t0 / 3) * 18 + p + t0 % 3) * 8, so1 = ((t1 / 3) * 18 + p + t1 % 3) * 8, so8 = (2 * 18 + p + 2) * 8;
  const bf16x4 z4 = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
  const int total = N * HW;                                      // < 2^31 (host)
  // Runs are dealt out grid-stride -- the block's four waves take four adjacent runs (8 KB of y), the next block the next four, and the whole grid moves on
  // together: at any time the chip writes one moving window of a few MB.
  const int step = (int)gridDim.x * 64;
  int q = ((int)blockIdx.x * 4 + wv) * 16;
  // (The prefetch is UNCONDITIONAL -- past the end it re-reads the last run: inside a branch, the compiler's wait-count bookkeeping merges the two paths
  // and waits for vmcnt(0) at the top of the loop, i.e. for this run's STORES to be acknowledged, every run.)
  bf16x4 din = z4;
  auto fetch = [&](int qq) {
    const int n = qq / HW, rem = qq - n * HW, yy = rem / W, x0 = rem - yy * W;
    if (l < 54) din = *reinterpret_cast<const bf16x4*>(xp + (((size_t)n * (H + 2) + yy + sr) * Wp + x0 + scol) * 4);
  };
  if (q >= total) return;
  fetch(q);
  for (; q < total; q += step) {
    if (l < 54) *reinterpret_cast<bf16x4*>(stage + l * 8) = din;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const bf16x4 d0 = *reinterpret_cast<const bf16x4*>(stage + so0), d1 = *reinterpret_cast<const bf16x4*>(stage + so1);
    const bf16x4 d8 = *reinterpret_cast<const bf16x4*>(stage + so8);      // every lane reads it; only k group 0's weights are non-zero there
    const bf16x8 b0 = __builtin_shufflevector(d0, d1, 0, 1, 2, 3, 4, 5, 6, 7);
    const bf16x8 b1 = __builtin_shufflevector(d8, z4, 0, 1, 2, 3, 4, 5, 6, 7);
    fetch(q + step < total ? q + step : q);
#pragma unroll
    for (int f = 0; f < NF; f++) {
      f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[f], b0, bs[f], 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[f], b1, acc, 0, 0, 0);
      bf16x4 o;
#pragma unroll
      for (int i = 0; i < 4; i++) o[i] = (bf16)(RELU ? fmaxf(acc[i], 0.f) : acc[i]);
      *reinterpret_cast<bf16x4*>(tile + p * ROW + (16 * f + 4 * kg) * 2) = o;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    bf16* yq = y + (size_t)q * COUT;
#pragma unroll
    for (int r = 0; r < ROUNDS; r++) {
      const int pix = r * PPI + l / CPP, c8 = l % CPP;
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(tile + pix * ROW + c8 * 16);
      __builtin_nontemporal_store(v, reinterpret_cast<bf16x8*>(yq + (size_t)pix * COUT + c8 * 8));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the tile is rewritten by the next run
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace dmvae_cin3
using namespace dmvae_cin3;

extern "C" int dmvae_conv_in3_supported(int n, int h, int w, int cout) {
  return (n > 0 && h > 0 && w > 0 && w % 16 == 0 && (cout == 64 || cout == 128) && (long long)n * h * w < (1ll << 31)) ? 1 : 0;
}
extern "C" size_t dmvae_conv_in3_workspace(int n, int h, int w) {
  if (n <= 0 || h <= 0 || w <= 0) return 0;
  return (size_t)n * (h + 2) * (w + 2) * 4 * sizeof(bf16);
}
extern "C" int dmvae_conv_in3(const void* x0, const void* x1, int n0, const void* shift, const void* scale, const void* w, const void* bias, void* y,
                              void* workspace, size_t workspace_bytes, int n, int h, int wd, int cout, int act, hipStream_t stream) {
  DMVAE_CHECK_ARG(x0 && w && y && workspace, "conv_in3: null pointer");
  DMVAE_CHECK_ARG(dmvae_conv_in3_supported(n, h, wd, cout), "conv_in3: unsupported shape n=%d h=%d w=%d cout=%d (w %% 16 == 0, cout 64 or 128)", n, h, wd, cout);
  DMVAE_CHECK_ARG(n0 >= 0 && n0 <= n && (n0 == n || x1), "conv_in3: %d of %d images in the first tensor but no second one", n0, n);
  DMVAE_CHECK_ARG((shift == nullptr) == (scale == nullptr), "conv_in3: shift and scale go together");
  DMVAE_CHECK_ARG(act == 0 || act == 2, "conv_in3: act must be 0 (none) or 2 (ReLU)");
  DMVAE_CHECK_ARG(workspace_bytes >= dmvae_conv_in3_workspace(n, h, wd), "conv_in3: workspace too small");
  bf16* xp = (bf16*)workspace;
  const size_t padded = (size_t)n * (h + 2) * (wd + 2);
  hipLaunchKernelGGL(pad_kernel, dim3((unsigned)((padded + 255) / 256)), dim3(256), 0, stream, (const float*)x0, (const float*)x1, n0, (const float*)shift,
                     (const float*)scale, xp, n, h, wd);
  DMVAE_CHECK_LAUNCH();
  const long long total = (long long)n * h * wd;
  const long long runs4 = (total + 63) / 64;      // groups of four runs
  const dim3 grid((unsigned)(runs4 < 2048 ? runs4 : 2048));      // 8 blocks per CU: every wave converts the weights once and then walks ~32 runs at 64 x 256 x 256
#define DMVAE_CIN3(NF, R) hipLaunchKernelGGL((conv_in3_kernel<NF, R>), grid, dim3(256), 0, stream, (const bf16*)xp, (const float*)w, (const float*)bias, (bf16*)y, n, h, wd)
  if (cout == 64) { if (act == 2) DMVAE_CIN3(4, true); else DMVAE_CIN3(4, false); }
  else { if (act == 2) DMVAE_CIN3(8, true); else DMVAE_CIN3(8, false); }
#undef DMVAE_CIN3
  DMVAE_CHECK_LAUNCH();
  return 0;
}
