// Implicit-GEMM convolution forward for NHWC bf16 activations on gfx950 (MI355X).
//
// Replaces the vendor conv the reference reaches through nn.Conv2d in
// models/flux_ae.py:63,65,67 (ResnetBlock), :32-35 (AttnBlock 1x1), :101 (Upsample conv,
// with the nearest-x2 of :104 folded into the gather), :274/:237 (conv_in / conv_out) and
// the nn.Linear GEMMs of models/vae.py:58-62 (ks=1, H=W=1).  The same kernel computes
// dgrad of a stride-1 conv when handed tap-flipped / transposed weights.
//
//   D[cout][pixel] = sum_{tap, ci} Wt[cout][tap][ci] * X[pixel (+) tap][ci]
//
// Tile 128 (cout) x 128 (pixels) x BK (ci chunk of one tap); 4 waves in 2x2, each wave a
// 64x64 sub-tile = 2x2 v_mfma_f32_32x32x16_bf16 accumulators.  Operand tiles are DMA'd
// straight into LDS with global_load_lds (16 B/lane); the XOR bank swizzle is applied on
// the per-lane *source* address (LDS image stays lane-linear) and again on the ds_read_b128
// fragment reads.  Padding pixels / out-of-range rows source a device zero page.
#include "common.h"
#include "dmvae_hip.h"
#include <cstdlib>

namespace dmvae_conv_fwd {

struct ConvArgs {
  const bf16* x;      // [N, Hi, Wi, Cin]
  const bf16* w;      // [Cout, T, Cin]
  const float* bias;  // [Cout] or null
  const bf16* res;    // [N, Ho, Wo, Cout] or null
  void* y;            // [N, Ho, Wo, Cout] bf16 or f32
  int N, Hi, Wi, Cin, Ho, Wo, Cout;
  int ks, act, M, ctiles, korder;
  // gather: tap (ky, kx) in [0, ks)^2 of output pixel (y, x) reads t = (y*so - pd + ky, x*so - pd + kx); sd = 1: source pixel t;
  // sd = 2, fl = 1: source t >> 1 (nearest x2 upsample folded in); sd = 2, fl = 0: source t / 2 for even t only (zero insertion: the
  // input gradient of a stride-2 conv); anything outside [0, Hi) x [0, Wi) reads zero
  int so, pd, sd, fl;
  long long x_bs, w_bs, y_bs;  // element strides per blockIdx.z (batched GEMM); 0 otherwise
};

constexpr int TM = 128;  // cout rows per tile  (MFMA "A" operand)
constexpr int TP = 128;  // pixel rows per tile (MFMA "B" operand)

template <int BK>
struct Geo {
  static constexpr int CPR = BK / 8;         // 16-B chunks per LDS row
  static constexpr int RPB = 16 / CPR;       // rows per 256-B bank row
  static constexpr int ROWB = BK * 2;        // bytes per LDS row
  static constexpr int TILEB = 128 * ROWB;   // bytes per operand tile
  static constexpr int LPW = TILEB / 1024 / 4;  // wave-loads per wave per tile
  __device__ static __forceinline__ int swz(int row) { return (row / RPB) % CPR; }
};

// TMk x TPk: cout x pixel tile.  128 x 128 (2 x 2 waves of 64 x 64) is the general shape; 32 x 256 (1 x 4 waves of 32 x 64, BK = 64 only)
// serves layers with a handful of output channels (conv_out 128 -> 3, the first VGG layer's input gradient 64 -> 3), where the 128-row
// tile spends 32x the useful matrix work on padding.
template <int BK, bool OUT_F32, int TMk = 128, int TPk = 128>
__global__ __launch_bounds__(256, 2) void conv_fwd_kernel(ConvArgs a) {
  using G = Geo<BK>;
  constexpr int WVM = TMk >= 128 ? 2 : 1, WVN = 4 / WVM;         // wave grid
  constexpr int IM = TMk / WVM / 32, JN = TPk / WVN / 32;          // 32 x 32 accumulator blocks per wave
  constexpr int WTILE = TMk * G::ROWB, PTILE = TPk * G::ROWB, STAGE = WTILE + PTILE;
  constexpr int LW = WTILE / 1024 / 4, LP = PTILE / 1024 / 4;      // wave-loads per wave per tile
  static_assert(LW >= 1 && LP >= 1 && IM >= 1 && JN >= 1, "tile too small for four waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // stage s: W tile at s*STAGE, P tile at s*STAGE + WTILE
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: the LDS-DMA destination (M0) derives from it -- as a VGPR value every issue became a readfirstlane waterfall loop
  const int wm = WVM == 2 ? wave >> 1 : 0, wn = WVM == 2 ? wave & 1 : wave;
  // flat grid, XCD-aware: cout tiles of one pixel tile are adjacent (they share the activation tile in L2) and each
  // XCD walks a contiguous range of pixel tiles (3x3 halo rows are shared in the same L2)
  const unsigned wid = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (int)(wid / a.ctiles) * TPk;   // first pixel
  const int n0 = (int)(wid % a.ctiles) * TMk;   // first cout
  const int T = a.ks * a.ks;
  const int nchunk = a.Cin / BK;
  const int S = T * nchunk;
  const bf16* zero = reinterpret_cast<const bf16*>(dmvae_zero_page);
  a.x += (size_t)blockIdx.z * a.x_bs;
  a.w += (size_t)blockIdx.z * a.w_bs;
  if (a.res) a.res += (size_t)blockIdx.z * a.y_bs;
  a.y = OUT_F32 ? (void*)(reinterpret_cast<float*>(a.y) + (size_t)blockIdx.z * a.y_bs)
                : (void*)(reinterpret_cast<bf16*>(a.y) + (size_t)blockIdx.z * a.y_bs);

  // --- per-thread load rows -------------------------------------------------------------
  int prow_n[LP], prow_y[LP], prow_x[LP];
  const bf16* wrow[LW];
  int csrcW[LW], csrcP[LP];  // source chunk (elements) for this lane in each of its rows
#pragma unroll
  for (int j = 0; j < LW; j++) {
    const int p = (wave * LW + j) * 64 + lane;
    const int row = p / G::CPR, cp = p % G::CPR;
    csrcW[j] = (cp ^ G::swz(row)) * 8;
    const int co = n0 + row;
    wrow[j] = co < a.Cout ? a.w + (size_t)co * T * a.Cin : nullptr;
  }
#pragma unroll
  for (int j = 0; j < LP; j++) {
    const int p = (wave * LP + j) * 64 + lane;
    const int row = p / G::CPR, cp = p % G::CPR;
    csrcP[j] = (cp ^ G::swz(row)) * 8;
    const int m = m0 + row;
    if (m < a.M) {
      const int hw = a.Ho * a.Wo;
      const int n = m / hw, r = m - n * hw;
      prow_n[j] = n; prow_y[j] = r / a.Wo; prow_x[j] = r - prow_y[j] * a.Wo;
    } else { prow_n[j] = -1; prow_y[j] = 0; prow_x[j] = 0; }
  }

  auto stage = [&](int s, int buf) {
    // channel chunk outer, tap inner: the 9 taps re-read (shifted) the same activation lines back to back -> L2 hits
    const int ch = a.korder ? s / T : s % nchunk, tap = a.korder ? s - ch * T : s / nchunk;
    const int ky = tap / a.ks, kx = tap - ky * a.ks;
    char* wt = smem + buf * STAGE;
    char* pt = wt + WTILE;
#pragma unroll
    for (int j = 0; j < LW; j++) {
      const int q = wave * LW + j;
      const bf16* src = wrow[j] ? wrow[j] + (size_t)tap * a.Cin + ch * BK + csrcW[j] : zero;
      __builtin_amdgcn_global_load_lds(GPTR(src), LPTR(wt + q * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < LP; j++) {
      const int q = wave * LP + j;
      const bf16* src = zero;
      int iy = prow_y[j] * a.so - a.pd + ky, ix = prow_x[j] * a.so - a.pd + kx;
      bool ok = prow_n[j] >= 0 && iy >= 0 && ix >= 0;
      if (a.sd == 2) {
        if (!a.fl) ok = ok && !((iy | ix) & 1);
        iy >>= 1; ix >>= 1;
      }
      if (ok && iy < a.Hi && ix < a.Wi) src = a.x + ((size_t)(prow_n[j] * a.Hi + iy) * a.Wi + ix) * a.Cin + ch * BK + csrcP[j];
      __builtin_amdgcn_global_load_lds(GPTR(src), LPTR(pt + q * 1024), 16, 0, 0);
    }
  };

  f32x16 acc[IM][JN];
#pragma unroll
  for (int i = 0; i < IM; i++)
#pragma unroll
    for (int j = 0; j < JN; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  // fragment read offsets (bytes within a tile), per kk step add chunk XOR
  int wro[IM], pro[JN], wsw[IM], psw[JN];
#pragma unroll
  for (int i = 0; i < IM; i++) {
    const int rw = wm * (IM * 32) + i * 32 + (lane & 31);
    wro[i] = rw * G::ROWB; wsw[i] = G::swz(rw);
  }
#pragma unroll
  for (int j = 0; j < JN; j++) {
    const int rp = wn * (JN * 32) + j * 32 + (lane & 31);
    pro[j] = rp * G::ROWB; psw[j] = G::swz(rp);
  }
  const int kg = lane >> 5;

  stage(0, 0);
  for (int s = 0; s < S; s++) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (s + 1 < S) stage(s + 1, (s + 1) & 1);
    const char* wt = smem + (s & 1) * STAGE;
    const char* pt = wt + WTILE;
#pragma unroll
    for (int kk = 0; kk < BK / 16; kk++) {
      const int c = kk * 2 + kg;
      bf16x8 wf[IM], pf[JN];
#pragma unroll
      for (int i = 0; i < IM; i++) wf[i] = *reinterpret_cast<const bf16x8*>(wt + wro[i] + ((c ^ wsw[i]) << 4));
#pragma unroll
      for (int j = 0; j < JN; j++) pf[j] = *reinterpret_cast<const bf16x8*>(pt + pro[j] + ((c ^ psw[j]) << 4));
#pragma unroll
      for (int i = 0; i < IM; i++)
#pragma unroll
        for (int j = 0; j < JN; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], pf[j], acc[i][j], 0, 0, 0);
    }
  }

  // --- epilogue: lane owns pixel (l&31) and 4-cout quads ---------------------------------
#pragma unroll
  for (int j = 0; j < JN; j++) {
    const int m = m0 + wn * (JN * 32) + j * 32 + (lane & 31);
    if (m >= a.M) continue;
#pragma unroll
    for (int i = 0; i < IM; i++) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int cb = n0 + wm * (IM * 32) + i * 32 + 8 * q + 4 * kg;
        if (cb >= a.Cout) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] = acc[i][j][4 * q + e];
        if (a.bias) {
          const f32x4 b = *reinterpret_cast<const f32x4*>(a.bias + cb);
#pragma unroll
          for (int e = 0; e < 4; e++) v[e] += b[e];
        }
        const size_t off = (size_t)m * a.Cout + cb;
        if (a.res) {
          const bf16x4 r = *reinterpret_cast<const bf16x4*>(a.res + off);
          if (a.act == 3) {  // ReLU-backward gate: `res` is the saved activation, not an addend
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = (float)r[e] > 0.f ? v[e] : 0.f;
          } else {
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] += (float)r[e];
          }
        }
        if (a.act == 1) {
#pragma unroll
          for (int e = 0; e < 4; e++) v[e] = v[e] * sigmoidf_(v[e]);
        } else if (a.act == 2) {
#pragma unroll
          for (int e = 0; e < 4; e++) v[e] = v[e] > 0.f ? v[e] : 0.f;
        } else if (a.act == 4) {  // LeakyReLU(0.2), models/patchgan.py:125
#pragma unroll
          for (int e = 0; e < 4; e++) v[e] = v[e] > 0.f ? v[e] : 0.2f * v[e];
        }
        if (OUT_F32) {
          f32x4 o = {v[0], v[1], v[2], v[3]};
          *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.y) + off) = o;
        } else {
          bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
          *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(a.y) + off) = o;
        }
      }
    }
  }
}

template <int BK, bool F32, int TMk = 128, int TPk = 128>
int launch(const ConvArgs& a_in, hipStream_t st, int batch = 1) {
  ConvArgs a = a_in;
  a.ctiles = (a.Cout + TMk - 1) / TMk;
  a.korder = 1;      // channel chunk outer, tap inner
  dim3 grid(((a.M + TPk - 1) / TPk) * a.ctiles, 1, batch);
  const int lds = 2 * (TMk + TPk) * Geo<BK>::ROWB;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fwd_kernel<BK, F32, TMk, TPk>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done = true;
  }
  hipLaunchKernelGGL((conv_fwd_kernel<BK, F32, TMk, TPk>), grid, dim3(256), lds, st, a);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

}  // namespace dmvae_conv_fwd
using namespace dmvae_conv_fwd;

int dmvae_conv_thin_try(const void* x, const void* w, const void* bias, const void* residual, void* y, const dmvae_conv_desc* d, hipStream_t stream);  // conv_thin.hip
int dmvae_conv_pp_try(const void* x, const void* w, const void* bias, const void* residual, void* y, const dmvae_conv_desc* d,
                      hipStream_t stream, float* gnpart, int gn_groups, int* gn_tp);  // conv_pp.hip; returns 1 when it declines the shape
int dmvae_gn_stats_from_quads(const float* part, float* stats, int n, int hw, int c, int groups, int tp, float eps, hipStream_t stream);  // groupnorm.hip

// Output size and gather parameters (see ConvArgs) of a conv descriptor; shared with conv_wgrad.hip.  Returns non-zero for combinations
// that are not built.
//   ks 1                      : so 1, pd 0
//   ks 3, stride 1            : so 1, pd 1  [upsample 1: nearest x2, sd 2 fl 1, output 2h x 2w]
//   ks 3, stride 2            : Downsample conv, input padded bottom/right only (flux_ae.py:85-95): so 2, pd 0, output h/2 x w/2
//   ks 3, upsample 2          : its input gradient = zero insertion: so 1, pd 2, sd 2, output 2h x 2w   (also: ks 3, stride 2, transposed)
//   ks 4, stride 1|2          : PatchGAN convs, padding 1 (patchgan.py:125-147): so = stride, pd 1, output (h + 2 - 4)/stride + 1
//   ks 4, stride 1|2, transposed : their input gradients: so 1, pd 2, sd = stride, output (h - 1)*stride + 2
int dmvae_conv_geometry(const dmvae_conv_desc* d, int* ho, int* wo, int* so, int* pd, int* sd, int* fl) {
  const int stride = d->stride == 2 ? 2 : 1;
  if (d->stride < 0 || d->stride > 2 || d->upsample < 0 || d->upsample > 2 || d->transposed < 0 || d->transposed > 1) return 1;
  *so = 1; *pd = 0; *sd = 1; *fl = 0; *ho = d->h; *wo = d->w;
  if (d->ks == 1) return (stride != 1 || d->upsample || d->transposed) ? 1 : 0;
  if (d->ks == 3) {
    *pd = 1;
    if (d->transposed) {
      if (stride != 2 || d->upsample) return 1;
      *pd = 2; *sd = 2; *ho = 2 * d->h; *wo = 2 * d->w;
      return 0;
    }
    if (stride == 2) {
      if (d->upsample || d->h % 2 || d->w % 2) return 1;
      *so = 2; *pd = 0; *ho = d->h / 2; *wo = d->w / 2;
      return 0;
    }
    if (d->upsample == 1) { *sd = 2; *fl = 1; *ho = 2 * d->h; *wo = 2 * d->w; }
    if (d->upsample == 2) { *pd = 2; *sd = 2; *ho = 2 * d->h; *wo = 2 * d->w; }
    return 0;
  }
  if (d->ks == 4) {
    if (d->upsample) return 1;
    if (d->transposed) { *pd = 2; *sd = stride; *ho = (d->h - 1) * stride + 2; *wo = (d->w - 1) * stride + 2; return 0; }
    if (d->h + 2 < 4 || d->w + 2 < 4) return 1;
    *so = stride; *pd = 1; *ho = (d->h + 2 - 4) / stride + 1; *wo = (d->w + 2 - 4) / stride + 1;
    return 0;
  }
  return 1;
}


extern "C" int dmvae_conv2d_nhwc_fwd(const void* x, const void* w, const void* bias, const void* residual,
                                     void* y, const dmvae_conv_desc* d, hipStream_t stream) {
  DMVAE_CHECK_ARG(x && w && y && d, "conv2d_nhwc_fwd: null pointer");
  DMVAE_CHECK_ARG(d->ks == 1 || d->ks == 3 || d->ks == 4, "conv2d_nhwc_fwd: ks must be 1, 3 or 4 (got %d)", d->ks);
  DMVAE_CHECK_ARG(d->cin > 0 && d->cin % 32 == 0, "conv2d_nhwc_fwd: Cin must be a positive multiple of 32 (got %d)", d->cin);
  DMVAE_CHECK_ARG(d->cout > 0 && d->cout % 4 == 0, "conv2d_nhwc_fwd: Cout must be a positive multiple of 4 (got %d)", d->cout);
  DMVAE_CHECK_ARG(d->n > 0 && d->h > 0 && d->w > 0, "conv2d_nhwc_fwd: empty shape");
  DMVAE_CHECK_ARG(d->act >= 0 && d->act <= 4 && (d->act != 3 || residual), "conv2d_nhwc_fwd: bad activation code %d", d->act);
  {
    const int r = dmvae_conv_pp_try(x, w, bias, residual, y, d, stream, nullptr, 0, nullptr);  // large shapes: the ping-pong kernel
    if (r <= 0) return r;
  }
  DMVAE_CHECK_ARG(d->w_layout == 0, "conv2d_nhwc_fwd: w_layout %d is only accepted where dmvae_conv_kmajor_applies(d) is 1", d->w_layout);
  {
    const int r = dmvae_conv_thin_try(x, w, bias, residual, y, d, stream);  // 3x3 to four f32 output channels: the halo-tile kernel
    if (r <= 0) return r;
  }
  ConvArgs a;
  a.x = (const bf16*)x; a.w = (const bf16*)w; a.bias = (const float*)bias; a.res = (const bf16*)residual; a.y = y;
  a.N = d->n; a.Hi = d->h; a.Wi = d->w; a.Cin = d->cin; a.Cout = d->cout;
  {
    int ho, wo;
    DMVAE_CHECK_ARG(dmvae_conv_geometry(d, &ho, &wo, &a.so, &a.pd, &a.sd, &a.fl) == 0,
                    "conv2d_nhwc_fwd: unsupported combination ks=%d stride=%d upsample=%d transposed=%d h=%d w=%d", d->ks, d->stride,
                    d->upsample, d->transposed, d->h, d->w);
    a.Ho = ho; a.Wo = wo;
  }
  a.ks = d->ks; a.act = d->act;
  a.x_bs = a.w_bs = a.y_bs = 0;
  const long long M = (long long)a.N * a.Ho * a.Wo;
  DMVAE_CHECK_ARG(M < (1ll << 31) / 4, "conv2d_nhwc_fwd: too many pixels");
  a.M = (int)M;
  const bool f32 = d->out_f32 != 0;
  constexpr bool narrow_ok = true;
  if (narrow_ok && a.Cout <= 32 && a.Cin % 64 == 0 && a.M >= 4096)   // a handful of output channels: 32-row cout tile
    return f32 ? launch<64, true, 32, 256>(a, stream) : launch<64, false, 32, 256>(a, stream);
  if (a.Cin % 64 == 0) return f32 ? launch<64, true>(a, stream) : launch<64, false>(a, stream);
  return f32 ? launch<32, true>(a, stream) : launch<32, false>(a, stream);
}

// conv2d_nhwc_fwd + the GroupNorm statistics of its bf16 result.  Large shapes: the ping-pong kernel's STATS instantiation sums the rounded results in its
// epilogue and a finishing kernel combines the per-tile partials; everything else: the conv, then the two-kernel statistics pass over the result.
extern "C" size_t dmvae_conv2d_nhwc_fwd_gnstats_workspace(const dmvae_conv_desc* d, int groups) {
  if (!d || groups <= 0) return 0;
  int ho, wo, g0, g1, g2, g3;
  if (dmvae_conv_geometry(d, &ho, &wo, &g0, &g1, &g2, &g3)) return 0;
  const size_t M = (size_t)d->n * ho * wo;
  const size_t quads = ((M + 255) / 256 + 8) * (size_t)d->cout * 2 * sizeof(float);   // [pixel tiles of >= 256][4][Cout / 4][2]
  const size_t two_pass = dmvae_groupnorm_workspace(d->n, ho * wo, d->cout, groups);
  return quads > two_pass ? quads : two_pass;
}

extern "C" int dmvae_conv2d_nhwc_fwd_gnstats(const void* x, const void* w, const void* bias, const void* residual, void* y, void* stats, void* workspace,
                                             size_t workspace_bytes, int groups, float eps, const dmvae_conv_desc* d, hipStream_t stream) {
  DMVAE_CHECK_ARG(x && w && y && stats && workspace && d, "conv2d_nhwc_fwd_gnstats: null pointer");
  DMVAE_CHECK_ARG(!d->out_f32, "conv2d_nhwc_fwd_gnstats: the statistics are those of the bf16 result (out_f32 must be 0)");
  int ho, wo, g0, g1, g2, g3;
  DMVAE_CHECK_ARG(dmvae_conv_geometry(d, &ho, &wo, &g0, &g1, &g2, &g3) == 0, "conv2d_nhwc_fwd_gnstats: unsupported conv descriptor");
  DMVAE_CHECK_ARG(workspace_bytes >= dmvae_conv2d_nhwc_fwd_gnstats_workspace(d, groups) && dmvae_groupnorm_workspace(d->n, ho * wo, d->cout, groups) > 0,
                  "conv2d_nhwc_fwd_gnstats: workspace too small or unsupported GroupNorm shape (n=%d hw=%d c=%d groups=%d)", d->n, ho * wo, d->cout, groups);
  constexpr bool fuse = true;
  if (fuse && d->ks != 0 && d->cin % 32 == 0 && d->cout % 4 == 0 && d->act >= 0 && d->act <= 4 && (d->act != 3 || residual)) {
    int tp = 0;
    const int r = dmvae_conv_pp_try(x, w, bias, residual, y, d, stream, (float*)workspace, groups, &tp);
    if (r < 0) return r;
    if (r == 0 && tp > 0) return dmvae_gn_stats_from_quads((const float*)workspace, (float*)stats, d->n, ho * wo, d->cout, groups, tp, eps, stream);
    if (r == 0) return dmvae_groupnorm_stats(y, stats, workspace, workspace_bytes, d->n, ho * wo, d->cout, groups, eps, stream);
  }
  const int r = dmvae_conv2d_nhwc_fwd(x, w, bias, residual, y, d, stream);
  if (r) return r;
  return dmvae_groupnorm_stats(y, stats, workspace, workspace_bytes, d->n, ho * wo, d->cout, groups, eps, stream);
}

// C[b][m][n] = act( sum_k A[b][m][k] * B[b][n][k] + bias[n] + R[b][m][n] ), all row-major bf16 (C bf16 or f32).
extern "C" int dmvae_gemm_nt_batched(const void* A, const void* B, const void* bias, const void* R, void* C, int M, int N,
                                     int K, int batch, long long a_bs, long long b_bs, long long c_bs, int act, int out_f32,
                                     hipStream_t stream) {
  DMVAE_CHECK_ARG(A && B && C, "gemm_nt_batched: null pointer");
  DMVAE_CHECK_ARG(M > 0 && N > 0 && N % 4 == 0 && K > 0 && K % 32 == 0 && batch > 0 && batch < 65536,
                  "gemm_nt_batched: need N%%4==0, K%%32==0 (M=%d N=%d K=%d batch=%d)", M, N, K, batch);
  ConvArgs a;
  a.x = (const bf16*)A; a.w = (const bf16*)B; a.bias = (const float*)bias; a.res = (const bf16*)R; a.y = C;
  a.N = 1; a.Hi = 1; a.Wi = M; a.Ho = 1; a.Wo = M; a.Cin = K; a.Cout = N; a.ks = 1; a.so = 1; a.pd = 0; a.sd = 1; a.fl = 0; a.act = act; a.M = M;
  a.x_bs = a_bs; a.w_bs = b_bs; a.y_bs = c_bs;
  if (K % 64 == 0) return out_f32 ? launch<64, true>(a, stream, batch) : launch<64, false>(a, stream, batch);
  return out_f32 ? launch<32, true>(a, stream, batch) : launch<32, false>(a, stream, batch);
}
