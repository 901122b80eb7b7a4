// 3x3 stride-1 convolution to FOUR output channels (f32 result) on gfx950 -- the decoder's conv_out (models/flux_ae.py:237,274: 128 -> 3 channels at
// 256 x 256, the image itself) and the input gradient of the LPIPS trunk's first layer (models/lpips.py:116-153, VGG conv1_1: 64 -> 3).
//
// With 4 of a matrix tile's 16 / 32 / 128 output rows in use the matrix pipe is idle whatever the tile; what the general kernel (conv_fwd.hip, 32 x 256
// tile) pays for is staging: every tap re-reads its pixel tile from L2 into LDS, 9 x the activation bytes (4.8 GB per conv_out call, 559 us).  This kernel
// is shaped by the bytes instead: a block owns 4 x 32 output pixels, stages the 6 x 34-pixel halo tile ONCE (1.6 x the tile's own pixels; neighbouring
// blocks' halos hit in L2) and takes all nine taps from it with a shifted pixel index.  HBM-bound: activation bytes once + 16 B per output pixel.
//
//   GEMM view per block:  D[cout 0..15][pixel] = sum_{tap, ci} W[cout][tap][ci] * X[pixel (+) tap][ci]      (v_mfma_f32_16x16x32_bf16, couts 4..15 are zeros)
//   - M side (A operand) = the weights: one fragment per (tap, 32-channel chunk), loop-invariant, in registers; the four waves split the 9 * Cin / 32
//     fragments between them (K split) and their partial sums meet in LDS.
//   - N side (B operand) = 16 consecutive pixels of an output row; lane l supplies pixel l & 15, channels 8 (l >> 4) .. + 7 of the chunk: one ds_read_b128.
//     The halo tile is pixel-linear in LDS as the LDS-DMA writes it; 16-B chunks are XOR-swizzled by the pixel index (on the DMA source address and on the
//     read) so that the 16 pixels of a lane group fall into 16 distinct bank slots.
//   - results: lanes 0..15 hold couts 0..3 of their pixel -- one 16-B store per pixel, 512 B contiguous per tile row.
// Zero padding = out-of-range buffer offsets (hardware zeros), as in conv_pp.hip.  No activation / residual (neither call site has one); bias optional.
#include "common.h"
#include "dmvae_hip.h"
#include <cstdlib>

namespace dmvae_conv_thin {

struct Args {
  const bf16* x;      // [N, H, W, CIN]
  const bf16* w;      // [4][9][CIN]
  const float* bias;  // [4] or null
  float* y;           // [N, H, W, 4]
  int N, H, W;
  int tiles_x, tiles_y;
  // NORM instantiation (the decoder's tail forward, flux_ae.py:266-268): x is norm_out's INPUT; the tile is normalised + swished on its way into LDS and its
  // interior is also written out as `aout` (what the weight gradient reads later); y is the NCHW f32 image with `cout` channels
  const float* stats; const float* gamma; const float* beta;
  bf16* aout;
  int G, cout;
  // nchw: y is [N][cout][H][W] f32 (cout <= 4 planes) instead of [N][H][W][4]; mul: optional per-output-channel multiplier applied last (device, [cout]) -- the
  // LPIPS trunk's image gradient leaves as the NCHW tensor autograd wants, already divided by the ScalingLayer's scale and multiplied by the incoming gradient
  int nchw;
  const float* mul;
};

constexpr unsigned SENT = 0x80000000u;

template <int CIN, bool NORM = false>
__global__ __launch_bounds__(256) void conv_thin_kernel(Args a) {
#if __HIP_DEVICE_COMPILE__
  constexpr int TR = 4, TC = 32;              // output tile
  constexpr int HC = TC + 2, NPX = (TR + 2) * HC;  // halo tile: 6 x 34 = 204 pixels
  constexpr int PIXB = CIN * 2;               // bytes per pixel
  constexpr int CPP = PIXB / 16;              // 16-B chunks per pixel: 16 (CIN 128) / 8 (CIN 64)
  constexpr int PPP = 1024 / PIXB;            // pixels per 1-KiB DMA piece
  constexpr int NPIECE = (NPX + PPP - 1) / PPP;
  constexpr int NCH = CIN / 32, NF = 9 * NCH; // (tap, channel chunk) fragments
  constexpr int FPW = (NF + 3) / 4;           // per wave
  static_assert(CIN == 64 || CIN == 128, "CIN");
  extern __shared__ __attribute__((aligned(16))) char smem[];   // NPIECE KiB halo tile; its first 8 KiB are reused for the waves' partial sums

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned blk = xcd_remap(blockIdx.x, gridDim.x);
  const int tpi = a.tiles_x * a.tiles_y;
  const int n = __builtin_amdgcn_readfirstlane((int)(blk / (unsigned)tpi));
  const int t = (int)blk - n * tpi;
  const int ty = __builtin_amdgcn_readfirstlane(t / a.tiles_x);
  const int y0 = ty * TR, x0 = (t - ty * a.tiles_x) * TC;

  // swizzle key of halo pixel q: CIN 128 -> 16 chunks of a pixel fill one 256-B bank row: key q & 15; CIN 64 -> two pixels per bank row: key (q >> 1) & 7
  auto key = [](int q) { return CPP == 16 ? (q & 15) : ((q >> 1) & 7); };

  // ---- stage the halo tile -------------------------------------------------------------------------------------------------------------------------
  const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (unsigned)a.N * a.H * a.W * PIXB, 0x00020000);
  const unsigned img = (unsigned)n * a.H * a.W * PIXB;
  constexpr int NJ = (NPX * CPP + 255) / 256;      // NORM: 16-B chunks per thread (13 at CIN = 128); thread <-> chunk tid % CPP of pixels tid / CPP + (256 / CPP) j
  if constexpr (NORM) {
    // through registers: GroupNorm + swish applied once per element on the way in (the halo's share, 1.6 x the tile, is recomputed by the neighbours), zero
    // padding applied AFTER the activation, the tile's own 4 x 32 pixels also stored as the activation tensor the backward reads.  Two halves, each with all of
    // its loads in flight first (one batch of 13 costs 52 registers next to the weight fragments and the accumulators: two waves per SIMD).
    const int cp = tid % CPP;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const int c = cp * 8 + e;
      const float* st = a.stats + ((size_t)n * a.G + c / (CIN / a.G)) * 2;
      sc[e] = st[1] * a.gamma[c];
      sh[e] = a.beta[c] - st[0] * sc[e];                  // apply_kernel's arithmetic (csrc/groupnorm.hip): the same bits as the stand-alone pass
    }
    constexpr int HALF = (NJ + 1) / 2;
#pragma unroll
    for (int h0 = 0; h0 < NJ; h0 += HALF) {
      bf16x8 xin[HALF];
#pragma unroll
      for (int u = 0; u < HALF; u++) {
        const int j = h0 + u;
        const int idx = tid + 256 * j, q = idx / CPP;
        const int r = q / HC, c = q - r * HC;
        const int yy = y0 - 1 + r, xx = x0 - 1 + c;
        const bool ok = j < NJ && q < NPX && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
        const unsigned vo = ok ? (unsigned)(yy * a.W + xx) * PIXB + (unsigned)(cp * 16) : SENT;
        xin[u] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rX, vo, img, 0));
      }
#pragma unroll
      for (int u = 0; u < HALF; u++) {
        const int j = h0 + u;
        const int idx = tid + 256 * j, q = idx / CPP;
        if (j >= NJ || q >= NPX) break;
        const int r = q / HC, c = q - r * HC;
        const int yy = y0 - 1 + r, xx = x0 - 1 + c;
        const bool ok = (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; e++) {
          const float t = (float)xin[u][e] * sc[e] + sh[e];
          o[e] = ok ? (bf16)(t * sigmoidf_(t)) : (bf16)0.f;
        }
        *reinterpret_cast<bf16x8*>(smem + q * PIXB + ((cp ^ key(q)) * 16)) = o;
        if (r >= 1 && r <= TR && c >= 1 && c <= TC) *reinterpret_cast<bf16x8*>(a.aout + ((size_t)(n * a.H + yy) * a.W + xx) * CIN + cp * 8) = o;
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < (NPIECE + 3) / 4; j++) {
      const int p = wave + 4 * j;
      if (p >= NPIECE) break;  // wave-uniform
      const int q = p * PPP + lane / CPP, cp = lane % CPP;
      const int r = q / HC, c = q - r * HC;
      const int yy = y0 - 1 + r, xx = x0 - 1 + c;
      const bool ok = q < NPX && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
      const unsigned vo = ok ? (unsigned)(yy * a.W + xx) * PIXB + (unsigned)((cp ^ key(q)) * 16) : SENT;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, LPTR(smem + p * 1024), 16, vo, img, 0, 0);
    }
  }

  // ---- this wave's weight fragments (couts >= 4: zeros) while the tile is in flight ------------------------------------------------------------------
  bf16x8 wf[FPW];
  const int co = lane & 15;
#pragma unroll
  for (int i = 0; i < FPW; i++) {
    const int f = wave + 4 * i;
    bf16x8 z;
#pragma unroll
    for (int e = 0; e < 8; e++) z[e] = (bf16)0.0f;
    wf[i] = z;
    if (f < NF && co < 4) {
      const int tap = f / NCH, ch = f - tap * NCH;
      wf[i] = *reinterpret_cast<const bf16x8*>(a.w + ((size_t)co * 9 + tap) * CIN + ch * 32 + (lane >> 4) * 8);
    }
  }
  f32x4 acc[8];
#pragma unroll
  for (int g = 0; g < 8; g++)
#pragma unroll
    for (int r = 0; r < 4; r++) acc[g][r] = 0.f;

  if constexpr (!NORM) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's pieces have landed
  __syncthreads();

  // ---- nine taps from the one tile ---------------------------------------------------------------------------------------------------------------------
#pragma unroll
  for (int i = 0; i < FPW; i++) {
    const int f = wave + 4 * i;
    if (f >= NF) break;  // wave-uniform
    const int tap = f / NCH, ch = f - tap * NCH;
    const int ky = tap / 3, kx = tap - ky * 3;
    const int cidx = ch * 4 + (lane >> 4);
#pragma unroll
    for (int g = 0; g < 8; g++) {
      const int q = ((g >> 1) + ky) * HC + (g & 1) * 16 + kx + (lane & 15);
      const bf16x8 b = *reinterpret_cast<const bf16x8*>(smem + q * PIXB + ((cidx ^ key(q)) * 16));
      acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], b, acc[g], 0, 0, 0);
    }
  }

  // ---- K split across the waves: partial sums through LDS (the halo tile is dead), then one 16-B store per pixel --------------------------------------
  __syncthreads();
  f32x4* part = reinterpret_cast<f32x4*>(smem);
  if (lane < 16) {
#pragma unroll
    for (int g = 0; g < 8; g++) part[wave * 128 + g * 16 + lane] = acc[g];
  }
  __syncthreads();
  if (tid < 128) {
    f32x4 s = part[tid];
#pragma unroll
    for (int w2 = 1; w2 < 4; w2++) s += part[w2 * 128 + tid];   // fixed order
    const int row = tid >> 5, col = tid & 31;
    if (NORM || a.nchw) {      // the image itself: NCHW f32, `cout` planes, 128 B contiguous per tile row and plane
#pragma unroll
      for (int co = 0; co < 4; co++)
        if (co < a.cout) {
          const float v = s[co] + (a.bias ? a.bias[co] : 0.f);
          a.y[((size_t)(n * a.cout + co) * a.H + y0 + row) * a.W + x0 + col] = a.mul ? v * a.mul[co] : v;
        }
    } else {
      if (a.bias) s += *reinterpret_cast<const f32x4*>(a.bias);
      *reinterpret_cast<f32x4*>(a.y + ((size_t)(n * a.H + y0 + row) * a.W + x0 + col) * 4) = s;
    }
  }
#endif
}

template <int CIN, bool NORM = false>
int launch(const Args& a, hipStream_t st) {
  constexpr int lds = ((6 * 34 + 1024 / (CIN * 2) - 1) / (1024 / (CIN * 2))) * 1024;
  static_assert(lds >= 8192, "partial sums reuse the tile");
  hipLaunchKernelGGL((conv_thin_kernel<CIN, NORM>), dim3((unsigned)(a.N * a.tiles_x * a.tiles_y)), dim3(256), lds, st, a);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

}  // namespace dmvae_conv_thin

// 0: launched; 1: not this kernel's shape (the caller goes on to the general kernel); < 0: error.
int dmvae_conv_thin_try(const void* x, const void* w, const void* bias, const void* residual, void* y, const dmvae_conv_desc* d, hipStream_t stream) {
  constexpr bool on = true;
  if (!on || d->ks != 3 || d->stride > 1 || d->upsample || d->transposed || !d->out_f32 || d->cout != 4 || (d->cin != 64 && d->cin != 128) || d->act != 0 ||
      residual || d->h % 4 != 0 || d->w % 32 != 0)
    return 1;
  if ((long long)d->n * d->h * d->w * d->cin * 2 >= (1ll << 31)) return 1;
  using namespace dmvae_conv_thin;
  Args a;
  a.x = (const bf16*)x; a.w = (const bf16*)w; a.bias = (const float*)bias; a.y = (float*)y;
  a.N = d->n; a.H = d->h; a.W = d->w; a.tiles_x = d->w / 32; a.tiles_y = d->h / 4;
  a.stats = a.gamma = a.beta = nullptr; a.aout = nullptr; a.G = 0; a.cout = 4; a.nchw = 0; a.mul = nullptr;
  return d->cin == 128 ? launch<128>(a, stream) : launch<64>(a, stream);
}

// ---- the decoder's tail forward, conv_out(swish(norm_out(x))) (flux_ae.py:266-268), in one launch: see the NORM members of Args ----
extern "C" int dmvae_norm_conv_out_fwd_supported(int n, int h, int w, int c, int groups, int cout) {
  return (n > 0 && h > 0 && w > 0 && c == 128 && groups > 0 && c % groups == 0 && cout >= 1 && cout <= 4 && h % 4 == 0 && w % 32 == 0 &&
          (long long)n * h * w * c * 2 < (1ll << 31)) ? 1 : 0;
}
extern "C" int dmvae_norm_conv_out_fwd(const void* x, const void* stats, const void* gamma, const void* beta, const void* w, const void* bias, void* aout, void* y,
                                       int n, int h, int wd, int c, int groups, int cout, hipStream_t stream) {
  DMVAE_CHECK_ARG(x && stats && gamma && beta && w && aout && y, "norm_conv_out_fwd: null pointer");
  DMVAE_CHECK_ARG(dmvae_norm_conv_out_fwd_supported(n, h, wd, c, groups, cout),
                  "norm_conv_out_fwd: unsupported shape n=%d h=%d w=%d c=%d groups=%d cout=%d (c = 128, h %% 4 == 0, w %% 32 == 0, cout <= 4)", n, h, wd, c, groups, cout);
  using namespace dmvae_conv_thin;
  Args a;
  a.x = (const bf16*)x; a.w = (const bf16*)w; a.bias = (const float*)bias; a.y = (float*)y;
  a.N = n; a.H = h; a.W = wd; a.tiles_x = wd / 32; a.tiles_y = h / 4;
  a.stats = (const float*)stats; a.gamma = (const float*)gamma; a.beta = (const float*)beta; a.aout = (bf16*)aout; a.G = groups; a.cout = cout; a.nchw = 1; a.mul = nullptr;
  return launch<128, true>(a, stream);
}


// 3x3 conv to cout <= 4 channels with the result as an NCHW f32 image, optionally scaled per channel: the input gradient of the LPIPS trunk's first layer
// (utils/lpips.py:81-104 backward: 64 -> 3 channels, then / ScalingLayer.scale and * the incoming gradient) in one launch instead of four.
extern "C" int dmvae_conv_to_image_supported(int n, int h, int w, int cin, int cout) {
  return (n > 0 && h > 0 && w > 0 && (cin == 64 || cin == 128) && cout >= 1 && cout <= 4 && h % 4 == 0 && w % 32 == 0 && (long long)n * h * w * cin * 2 < (1ll << 31)) ? 1 : 0;
}
extern "C" int dmvae_conv_to_image(const void* x, const void* w, const void* bias, const void* mul, void* y, int n, int h, int wd, int cin, int cout, hipStream_t stream) {
  DMVAE_CHECK_ARG(x && w && y, "conv_to_image: null pointer");
  DMVAE_CHECK_ARG(dmvae_conv_to_image_supported(n, h, wd, cin, cout), "conv_to_image: unsupported shape n=%d h=%d w=%d cin=%d cout=%d (cin 64 / 128, h %% 4 == 0, w %% 32 == 0, cout <= 4)",
                  n, h, wd, cin, cout);
  using namespace dmvae_conv_thin;
  Args a;
  a.x = (const bf16*)x; a.w = (const bf16*)w; a.bias = (const float*)bias; a.y = (float*)y;
  a.N = n; a.H = h; a.W = wd; a.tiles_x = wd / 32; a.tiles_y = h / 4;
  a.stats = a.gamma = a.beta = nullptr; a.aout = nullptr; a.G = 0; a.cout = cout; a.nchw = 1; a.mul = (const float*)mul;
  return cin == 128 ? launch<128>(a, stream) : launch<64>(a, stream);
}
