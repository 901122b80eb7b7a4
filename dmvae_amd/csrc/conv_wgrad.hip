// Weight-gradient of the NHWC bf16 convolution on gfx950 (MI355X).
//
// Replaces the cuDNN wgrad autograd reaches for nn.Conv2d at models/flux_ae.py:32-35,63,65,67,
// 101,237,274 (and nn.Linear at models/vae.py:58-62 with ks=1).
//
//   dW[co][tap][ci] = sum_p dy[p][co] * a[p (+) tap][ci]        (reduction over pixels)
//
// Both operands are channel-contiguous in HBM while the reduction runs over pixels, so the MFMA
// fragments (8 consecutive k per lane) are produced with the gfx950 LDS transpose read
// ds_read_b64_tr_b16: LDS holds [pixel][channel] tiles exactly as DMA'd (global_load_lds, 16 B
// per lane), and each 16-lane group reads a [4 pixels][16 channels] block transposed.
// The 64-B segment XOR swizzle (segment ^= pixel&3) is applied on the DMA source address and
// on the transpose read, making the four pixel rows of a block hit distinct bank quarters.
//
// Split-K over pixel ranges: each workgroup owns one (cout-tile, cin-tile, tap) and a pixel
// range, and writes an f32 slab; dmvae_conv_wgrad's second kernel reduces the slabs in a fixed
// order (deterministic, no float atomics) into the PyTorch weight layout [cout][cin][kh][kw].
#include "common.h"
#include "dmvae_hip.h"
#include <cstdlib>

namespace dmvae_conv_wgrad {

struct WgradArgs {
  const bf16* dy;  // [M, Cout]
  const bf16* a;   // [N, Hi, Wi, Cin]
  float* slab;     // [splits][Cout][T][Cin]
  int N, Hi, Wi, Cin, Ho, Wo, Cout;
  int ks, M, kchunk;       // kchunk: pixels per split (multiple of 64)
  int so, pd, sd, fl;      // gather of the conv being differentiated, as in conv_fwd.hip::ConvArgs (forward kinds only)
  int ntiles;              // (cout-tile, cin-tile, tap) combinations per pixel range
  long long dy_bs, a_bs, slab_bs;  // per blockIdx.z element strides (batched TN GEMM); 0 otherwise
};

constexpr int BKP = 64;              // pixels per K step
constexpr int TILEB = BKP * 256;     // bytes per operand tile ([64][128] bf16)

__device__ __forceinline__ s16x4 tr_read(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}

__global__ __launch_bounds__(256, 2) void wgrad_kernel(WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: the LDS-DMA destination (M0) derives from it -- as a VGPR value every issue became a readfirstlane waterfall loop
  const int wm = wave >> 1, wn = wave & 1;
  const int T = a.ks * a.ks;
  const int ci_tiles = (a.Cin + 127) / 128;
  // flat grid, XCD-aware: all (cout-tile, cin-tile, tap) blocks of one pixel range are adjacent in dispatch order and
  // each XCD owns a contiguous run of pixel ranges, so the dy / activation tiles are fetched into one L2 once
  const unsigned wid = xcd_remap(blockIdx.x, gridDim.x);
  const int split = (int)(wid / a.ntiles);
  int t = (int)(wid % a.ntiles);
  const int tap = t % T; t /= T;
  const int ci0 = (t % ci_tiles) * 128;
  const int co0 = (t / ci_tiles) * 128;
  const int ky = tap / a.ks, kx = tap - ky * a.ks;
  const int k0 = split * a.kchunk;
  const int k1 = min(k0 + a.kchunk, a.M);
  const int S = (k1 - k0 + BKP - 1) / BKP;
  const bf16* zero = reinterpret_cast<const bf16*>(dmvae_zero_page);
  a.dy += (size_t)blockIdx.z * a.dy_bs;
  a.a += (size_t)blockIdx.z * a.a_bs;
  a.slab += (size_t)blockIdx.z * a.slab_bs;

  // per-thread load rows: load q = wave*4+j covers pixel rows 4q..4q+3, lane -> row 4q+lane/16,
  // physical chunk lane%16; logical chunk = swizzle^-1 (an involution)
  int pn[4], py[4], px[4];
  const int cphys = lane & 15;
  int clog[4];
  bool co_ok[4], ci_ok[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int row = (wave * 4 + j) * 4 + (lane >> 4);
    clog[j] = ((((cphys >> 2) ^ (row & 3)) << 2) | (cphys & 3)) * 8;  // element offset in the 128-ch row
    co_ok[j] = co0 + clog[j] < a.Cout;
    ci_ok[j] = ci0 + clog[j] < a.Cin;
    const int p = k0 + row;
    const int hw = a.Ho * a.Wo;
    const int n = p / hw, r = p - n * hw;
    pn[j] = n; py[j] = r / a.Wo; px[j] = r - py[j] * a.Wo;
  }

  auto stage = [&](int s, int buf) {
    char* dt = smem + buf * 2 * TILEB;
    char* at = dt + TILEB;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int q = wave * 4 + j;
      const int p = k0 + s * BKP + q * 4 + (lane >> 4);
      const bf16* src = (p < k1 && co_ok[j]) ? a.dy + (size_t)p * a.Cout + co0 + clog[j] : zero;
      __builtin_amdgcn_global_load_lds(GPTR(src), LPTR(dt + q * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int q = wave * 4 + j;
      const int p = k0 + s * BKP + q * 4 + (lane >> 4);
      const bf16* src = zero;
      int iy = py[j] * a.so - a.pd + ky, ix = px[j] * a.so - a.pd + kx;
      bool ok = p < k1 && ci_ok[j] && iy >= 0 && ix >= 0;
      if (a.sd == 2) { iy >>= 1; ix >>= 1; }   // nearest x2 upsample folded in (the only sd = 2 kind that has a weight gradient here)
      if (ok && iy < a.Hi && ix < a.Wi) src = a.a + ((size_t)(pn[j] * a.Hi + iy) * a.Wi + ix) * a.Cin + ci0 + clog[j];
      __builtin_amdgcn_global_load_lds(GPTR(src), LPTR(at + q * 1024), 16, 0, 0);
      // advance this row by one K step (64 pixels)
      px[j] += BKP;
      while (px[j] >= a.Wo) { px[j] -= a.Wo; if (++py[j] == a.Ho) { py[j] = 0; pn[j]++; } }
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  // transpose-read addressing: lane supplies the address of 4 contiguous channels of one pixel row
  const int g = (lane >> 4) & 1, kq = lane >> 5, rr = (lane & 15) >> 2, qq = lane & 3;
  int choff_d[2], choff_a[2];  // byte offset inside the 64-B segment + segment index
  int seg_d[2], seg_a[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int chd = wm * 64 + i * 32 + 16 * g + 4 * qq;
    const int cha = wn * 64 + i * 32 + 16 * g + 4 * qq;
    seg_d[i] = chd >> 5; choff_d[i] = (chd & 31) * 2;
    seg_a[i] = cha >> 5; choff_a[i] = (cha & 31) * 2;
  }

  if (S > 0) stage(0, 0);
  for (int s = 0; s < S; s++) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (s + 1 < S) stage(s + 1, (s + 1) & 1);
    const char* dt = smem + (s & 1) * 2 * TILEB;
    const char* at = dt + TILEB;
#pragma unroll
    for (int kk = 0; kk < BKP / 16; kk++) {
      union { bf16x8 v; s16x4 h[2]; } df[2], af[2];
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int row = kk * 16 + kq * 8 + h * 4 + rr;  // row & 3 == rr
#pragma unroll
        for (int i = 0; i < 2; i++) {
          df[i].h[h] = tr_read(dt + row * 256 + ((seg_d[i] ^ rr) << 6) + choff_d[i]);
          af[i].h[h] = tr_read(at + row * 256 + ((seg_a[i] ^ rr) << 6) + choff_a[i]);
        }
      }
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(df[i].v, af[j].v, acc[i][j], 0, 0, 0);
    }
  }

  // slab store: lane owns ci = (l&31), 16 couts per block
  float* slab = a.slab + (size_t)split * a.Cout * T * a.Cin;
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int ci = ci0 + wn * 64 + j * 32 + (lane & 31);
    if (ci >= a.Cin) continue;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int co = co0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kq;
        if (co < a.Cout) slab[((size_t)co * T + tap) * a.Cin + ci] = acc[i][j][r];
      }
  }
}

// out[co][ci][tap] (PyTorch [cout][cin][kh][kw]) = (accumulate ? out : 0) + sum_s slab[s][co][tap][ci]
// batched TN GEMM epilogue: out[b][co][ci] (bf16 or f32) = alpha * sum_s slab[b][s][co][ci]
template <typename OutT>
__global__ void tn_reduce_kernel(const float* __restrict__ slab, OutT* __restrict__ out, int splits, size_t total,
                                 long long slab_bs, long long out_bs, float alpha) {
  slab += (size_t)blockIdx.y * slab_bs;
  out += (size_t)blockIdx.y * out_bs;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < splits; k++) s += slab[(size_t)k * total + i];
    out[i] = (OutT)(alpha * s);
  }
}

// Blocks [0, rb) reduce the weight slabs; blocks [rb, gridDim.x), present when the ping-pong kernel left per-split column sums of dy behind the
// slabs, do colsum_final_kernel's work (same summation order, bit-identical) -- one launch instead of two per weight gradient.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out, int splits, int Cout, int T,
                                                           int Cin, int accumulate, int rb, const float* __restrict__ bpart,
                                                           float* __restrict__ dbias, int nparts) {
  if ((int)blockIdx.x >= rb) {
    __shared__ float sh[4][64];
    const int c = ((int)blockIdx.x - rb) * 64 + (threadIdx.x & 63), kl = threadIdx.x >> 6;
    float s = 0.f;
    if (c < Cout)
      for (int k = kl; k < nparts; k += 4) s += bpart[(size_t)k * Cout + c];
    sh[kl][threadIdx.x & 63] = s;
    __syncthreads();
    if (kl == 0 && c < Cout) {
      const float t = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
      dbias[c] = accumulate ? dbias[c] + t : t;
    }
    return;
  }
  // Slab block = (cout, 64 input channels): [T][64] of every slab is read as 16-B pieces (a 16-lane group per tap row), summed over the splits in order,
  // turned through LDS and written as the contiguous [64][T] run of PyTorch's [cout][cin][kh][kw] layout.  The first form walked the flat index with 64-bit
  // divisions per element and stored 4-B words T * 4 bytes apart (32 us per call on average, 1.35 ms per step).
  __shared__ float tile[16][68];
  const int cchunks = (Cin + 63) >> 6;
  const int co = (int)blockIdx.x / cchunks, ci0 = ((int)blockIdx.x - co * cchunks) * 64;
  const size_t total = (size_t)Cout * T * Cin;
  {
    const int tap = threadIdx.x >> 4, c = (threadIdx.x & 15) * 4;
    if (tap < T && ci0 + c < Cin) {   // Cin % 4 == 0 (host)
      const float* src = slab + ((size_t)co * T + tap) * Cin + ci0 + c;
      f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
      for (int k = 0; k < splits; k++) sum += *reinterpret_cast<const f32x4*>(src + (size_t)k * total);
      *reinterpret_cast<f32x4*>(&tile[tap][c]) = sum;
    }
  }
  __syncthreads();
  const int ncol = min(64, Cin - ci0), nout = ncol * T;
  const unsigned rcp = (65536u + (unsigned)T - 1u) / (unsigned)T;   // e / T for e < 1024, T <= 16
  float* dst = out + ((size_t)co * Cin + ci0) * T;
  for (int e = threadIdx.x; e < nout; e += 256) {
    const int c = (int)(((unsigned)e * rcp) >> 16), tap = e - c * T;
    const float v = tile[tap][c];
    dst[e] = accumulate ? dst[e] + v : v;
  }
}

// T > 1, one block per (cout, up to 512 input channels): [T][512] of every slab as 16-B pieces with ALL threads loading (the 64-channel form above keeps T * 16 of 256
// threads busy -- 144 at T = 9 -- and runs 8 x as many blocks of 2.3 KB each), summed over the splits in order, turned through LDS, written as the contiguous [512][T] run.
__global__ __launch_bounds__(256) void wgrad_reduce_row_kernel(const float* __restrict__ slab, float* __restrict__ out, int splits, int Cout, int T, int Cin,
                                                               int accumulate, int rb, int cchunks, const float* __restrict__ bpart, float* __restrict__ dbias,
                                                               int nparts) {
  if ((int)blockIdx.x >= rb) {
    __shared__ float sh[4][64];
    const int c = ((int)blockIdx.x - rb) * 64 + (threadIdx.x & 63), kl = threadIdx.x >> 6;
    float s = 0.f;
    if (c < Cout)
      for (int k = kl; k < nparts; k += 4) s += bpart[(size_t)k * Cout + c];
    sh[kl][threadIdx.x & 63] = s;
    __syncthreads();
    if (kl == 0 && c < Cout) {
      const float t = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
      dbias[c] = accumulate ? dbias[c] + t : t;
    }
    return;
  }
  __shared__ float tile[16][516];
  const int co = (int)blockIdx.x / cchunks, ci0 = ((int)blockIdx.x - co * cchunks) * 512;
  const int ncol = min(512, Cin - ci0), q4 = ncol >> 2;   // Cin % 4 == 0 (host)
  const size_t total = (size_t)Cout * T * Cin;
  for (int e = threadIdx.x; e < T * q4; e += 256) {
    const int tap = e / q4, c = (e - tap * q4) * 4;
    const float* src = slab + ((size_t)co * T + tap) * Cin + ci0 + c;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int k = 0; k < splits; k++) sum += *reinterpret_cast<const f32x4*>(src + (size_t)k * total);
    *reinterpret_cast<f32x4*>(&tile[tap][c]) = sum;
  }
  __syncthreads();
  const int nout = ncol * T;
  const unsigned rcp = (65536u + (unsigned)T - 1u) / (unsigned)T;   // e / T for e < 8192, T <= 16: exact ((e * rcp) >> 16 with e * rcp < 2^32 and error < 1 / T)
  float* dst = out + ((size_t)co * Cin + ci0) * T;
  for (int e = threadIdx.x; e < nout; e += 256) {
    const int c = (int)(((unsigned)e * rcp) >> 16), tap = e - c * T;
    const float v = tile[tap][c];
    dst[e] = accumulate ? dst[e] + v : v;
  }
}

// T == 1 (Linear layers, 1x1 convs): the slab layout [cout][1][cin] IS the output layout, so the reduce is a flat sum of `splits` vectors -- 16 B per lane, all lanes
// busy.  The tap-transposing kernel above keeps 16 of its 256 threads loading when T = 1 (one 256-B row per block and split: 40 us per call on LightningDiT's
// Linear weight gradients, 1.6 TB/s).  Same order of additions (split 0, 1, ...), same bias blocks behind the slab blocks.
__global__ __launch_bounds__(256) void wgrad_reduce_flat_kernel(const float* __restrict__ slab, float* __restrict__ out, int splits, size_t total4, int accumulate,
                                                                int rb, int Cout, const float* __restrict__ bpart, float* __restrict__ dbias, int nparts) {
  if ((int)blockIdx.x >= rb) {
    __shared__ float sh[4][64];
    const int c = ((int)blockIdx.x - rb) * 64 + (threadIdx.x & 63), kl = threadIdx.x >> 6;
    float s = 0.f;
    if (c < Cout)
      for (int k = kl; k < nparts; k += 4) s += bpart[(size_t)k * Cout + c];
    sh[kl][threadIdx.x & 63] = s;
    __syncthreads();
    if (kl == 0 && c < Cout) {
      const float t = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
      dbias[c] = accumulate ? dbias[c] + t : t;
    }
    return;
  }
  const f32x4* src = reinterpret_cast<const f32x4*>(slab);
  f32x4* dst = reinterpret_cast<f32x4*>(out);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)rb * 256) {
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int k = 0; k < splits; k++) sum += src[(size_t)k * total4 + i];
    dst[i] = accumulate ? dst[i] + sum : sum;
  }
}

// db[c] (+)= sum_p dy[p][c].  HBM-bound: 16-B loads (8 channels per lane), tp channel-lanes x (256/tp) pixel rows per
// block iteration, f32 accumulation, LDS block reduce, fixed-order second stage (deterministic).
__global__ __launch_bounds__(256) void colsum_partial_kernel(const bf16* __restrict__ dy, float* __restrict__ part, int M, int C,
                                                             int tp_shift, int rows_per_block) {
  __shared__ float red[256 * 8];
  const int tp = 1 << tp_shift, rows = 256 >> tp_shift;
  const int lc = threadIdx.x & (tp - 1), prow = threadIdx.x >> tp_shift;
  const int c0 = (blockIdx.y * tp + lc) * 8;
  const int p0 = blockIdx.x * rows_per_block, p1 = min(p0 + rows_per_block, M);
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (c0 < C)
    for (int p = p0 + prow; p < p1; p += rows) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(dy + (size_t)p * C + c0);
#pragma unroll
      for (int e = 0; e < 8; e++) s[e] += (float)v[e];
    }
#pragma unroll
  for (int e = 0; e < 8; e++) red[(prow * tp + lc) * 8 + e] = s[e];
  __syncthreads();
  for (int c = threadIdx.x; c < tp * 8; c += 256) {
    float a = 0.f;
    for (int r = 0; r < rows; r++) a += red[(r * tp + (c >> 3)) * 8 + (c & 7)];
    const int cg = blockIdx.y * tp * 8 + c;
    if (cg < C) part[(size_t)blockIdx.x * C + cg] = a;
  }
}
// 64 channels per block, 4 lanes per channel stride over the partials
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, float* __restrict__ out, int nparts, int C,
                                                           int accumulate) {
  __shared__ float sh[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), kl = threadIdx.x >> 6;
  float s = 0.f;
  if (c < C)
    for (int k = kl; k < nparts; k += 4) s += part[(size_t)k * C + c];
  sh[kl][threadIdx.x & 63] = s;
  __syncthreads();
  if (kl == 0 && c < C) {
    const float t = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
    out[c] = accumulate ? out[c] + t : t;
  }
}

static int colsum_launch(const bf16* dy, float* dbias, float* part, int M, int C, int accumulate, hipStream_t stream) {
  int tp = 1, tps = 0;
  while (tp < 64 && tp * 8 < C) { tp <<= 1; tps++; }
  const int rows = 256 / tp;
  int nparts = (M + rows * 8 - 1) / (rows * 8);   // >= 8 row iterations per block
  if (nparts > 512) nparts = 512;
  if (nparts < 1) nparts = 1;
  const int rpb = (M + nparts - 1) / nparts;
  nparts = (M + rpb - 1) / rpb;
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(nparts, (C + tp * 8 - 1) / (tp * 8)), dim3(256), 0, stream, dy, part, M, C, tps, rpb);
  DMVAE_CHECK_LAUNCH();
  hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 63) / 64), dim3(256), 0, stream, part, dbias, nparts, C, accumulate);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

static int pick_splits(int M, int tiles) {
  // aim for ~1024 workgroups, at least 8 K steps each
  int want = (1024 + tiles - 1) / tiles;
  int maxs = (M + 8 * BKP - 1) / (8 * BKP);
  int s = want < maxs ? want : maxs;
  return s < 1 ? 1 : s;
}

}  // namespace dmvae_conv_wgrad
using namespace dmvae_conv_wgrad;

// conv_wgrad_pp.hip: the ping-pong kernel for the large layers (plan returns 0 when it does not cover the shape)
int dmvae_wgrad_pp_plan(const dmvae_conv_desc* d, int* splits_out, int* kchunk_out, int* cfg_out);
int dmvae_conv_geometry(const dmvae_conv_desc* d, int* ho, int* wo, int* so, int* pd, int* sd, int* fl);  // conv_fwd.hip
int dmvae_wgrad_pp_launch(const void* dy, const void* act, float* slab, float* bslab, const dmvae_conv_desc* d, int splits, int kchunk,
                          int cfg, hipStream_t stream);

extern "C" size_t dmvae_conv2d_nhwc_wgrad_workspace(const dmvae_conv_desc* d) {
  if (!d) return 0;
  const int T = d->ks * d->ks;
  {
    int sp, kc, cfg;
    if (dmvae_wgrad_pp_plan(d, &sp, &kc, &cfg))
      return (size_t)sp * d->cout * T * d->cin * sizeof(float) + (size_t)4096 * d->cout * sizeof(float);  // + [splits*ntiles <= 4096][cout] bias partials
  }
  int ho = 0, wo = 0, g0, g1, g2, g3;
  if (dmvae_conv_geometry(d, &ho, &wo, &g0, &g1, &g2, &g3)) return 0;
  const long long M = (long long)d->n * ho * wo;
  const int tiles = ((d->cout + 127) / 128) * ((d->cin + 127) / 128) * T;
  const int splits = pick_splits((int)M, tiles);
  size_t slab = (size_t)splits * d->cout * T * d->cin * sizeof(float);
  size_t colsum = (size_t)1024 * d->cout * sizeof(float);
  return slab + colsum;
}

extern "C" int dmvae_conv2d_nhwc_wgrad(const void* dy, const void* a, void* dw, void* dbias, void* workspace,
                                       size_t workspace_bytes, const dmvae_conv_desc* d, int accumulate,
                                       hipStream_t stream) {
  DMVAE_CHECK_ARG(dy && a && dw && d && workspace, "conv2d_nhwc_wgrad: null pointer");
  DMVAE_CHECK_ARG(d->ks == 1 || d->ks == 3 || d->ks == 4, "conv2d_nhwc_wgrad: ks must be 1, 3 or 4");
  DMVAE_CHECK_ARG(d->cin > 0 && d->cin % 8 == 0 && d->cout > 0 && d->cout % 8 == 0,
                  "conv2d_nhwc_wgrad: Cin and Cout must be positive multiples of 8 (got %d, %d)", d->cin, d->cout);
  DMVAE_CHECK_ARG(workspace_bytes >= dmvae_conv2d_nhwc_wgrad_workspace(d), "conv2d_nhwc_wgrad: workspace too small");
  WgradArgs w;
  w.dy = (const bf16*)dy; w.a = (const bf16*)a; w.slab = (float*)workspace;
  w.N = d->n; w.Hi = d->h; w.Wi = d->w; w.Cin = d->cin; w.Cout = d->cout; w.ks = d->ks;
  DMVAE_CHECK_ARG(!d->transposed && d->upsample != 2 && dmvae_conv_geometry(d, &w.Ho, &w.Wo, &w.so, &w.pd, &w.sd, &w.fl) == 0,
                  "conv2d_nhwc_wgrad: unsupported combination ks=%d stride=%d upsample=%d transposed=%d h=%d w=%d", d->ks, d->stride,
                  d->upsample, d->transposed, d->h, d->w);
  const long long M = (long long)w.N * w.Ho * w.Wo;
  DMVAE_CHECK_ARG(M > 0 && M < (1ll << 31) / 4, "conv2d_nhwc_wgrad: bad pixel count");
  w.M = (int)M;
  w.dy_bs = w.a_bs = w.slab_bs = 0;
  const int T = w.ks * w.ks;
  const int tiles = ((w.Cout + 127) / 128) * ((w.Cin + 127) / 128) * T;
  int splits = pick_splits(w.M, tiles);
  int pp_kchunk = 0, pp_cfg = 0, pp_ntiles = 1;
  bool bias_fused = false;
  if (dmvae_wgrad_pp_plan(d, &splits, &pp_kchunk, &pp_cfg)) {
    const size_t tot = (size_t)w.Cout * T * w.Cin;
    const int rc = dmvae_wgrad_pp_launch(dy, a, w.slab, dbias ? w.slab + (size_t)splits * tot : nullptr, d, splits, pp_kchunk, pp_cfg, stream);
    if (rc) return rc;
    bias_fused = dbias != nullptr;
    { const int ng = T * (d->cin / 128); pp_ntiles = pp_cfg == 0 ? (ng + 1) / 2 : (ng + 2) / 3; }
  } else {
    w.kchunk = (((w.M + splits - 1) / splits) + BKP - 1) / BKP * BKP;
    splits = (w.M + w.kchunk - 1) / w.kchunk;
    static bool attr_done = false;
    if (!attr_done) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILEB);
      attr_done = true;
    }
    w.ntiles = tiles;
    hipLaunchKernelGGL(wgrad_kernel, dim3(splits * tiles), dim3(256), 4 * TILEB, stream, w);
    DMVAE_CHECK_LAUNCH();
  }
  const size_t total = (size_t)w.Cout * T * w.Cin;
  const int nb = bias_fused ? (w.Cout + 63) / 64 : 0;  // the ping-pong kernel left per-split column sums of dy behind the weight slabs
  constexpr bool flat_ok = true;
  if (flat_ok && T == 1 && total % 4 == 0) {
    const size_t total4 = total / 4;
    const int rbf = (int)((total4 + 255) / 256 < 2048 ? (total4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(wgrad_reduce_flat_kernel, dim3(rbf + nb), dim3(256), 0, stream, w.slab, (float*)dw, splits, total4, accumulate, rbf, w.Cout,
                       w.slab + (size_t)splits * total, (float*)dbias, splits * pp_ntiles);
  } else if (flat_ok && (T == 4 || T == 9 || T == 16) && w.Cin % 4 == 0 && w.Cin >= 128) {   // the reciprocal index split below is exact for these T
    const int cchunks = (w.Cin + 511) / 512, rb = w.Cout * cchunks;   // one block per (cout, up to 512 input channels)
    hipLaunchKernelGGL(wgrad_reduce_row_kernel, dim3(rb + nb), dim3(256), 0, stream, w.slab, (float*)dw, splits, w.Cout, T, w.Cin, accumulate, rb, cchunks,
                       w.slab + (size_t)splits * total, (float*)dbias, splits * pp_ntiles);
  } else {
    const int rb = w.Cout * ((w.Cin + 63) / 64);   // one block per (cout, 64 input channels)
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rb + nb), dim3(256), 0, stream, w.slab, (float*)dw, splits, w.Cout, T, w.Cin, accumulate, rb,
                       w.slab + (size_t)splits * total, (float*)dbias, splits * pp_ntiles);
  }
  DMVAE_CHECK_LAUNCH();
  if (!bias_fused && dbias) return colsum_launch(w.dy, (float*)dbias, w.slab + (size_t)splits * total, w.M, w.Cout, accumulate, stream);
  return 0;
}

// the same reduction for thousands of partial rows (one per GroupNorm chunk): 16 channels per block, 64 strided partial-lanes per channel with four loads in
// flight each, fixed-order LDS tree -- colsum_final_kernel's 4 lanes per channel walk 2048 rows in 512 dependent steps (83 us)
__global__ __launch_bounds__(1024) void colsum_final_wide_kernel(const float* __restrict__ part, float* __restrict__ out, int nparts, int C, int accumulate) {
  __shared__ float sh[64][17];
  const int cl = threadIdx.x & 15, kl = threadIdx.x >> 4, c = blockIdx.x * 16 + cl;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (c < C) {
    int k = kl;
    for (; k + 192 < nparts; k += 256) {
      a0 += part[(size_t)k * C + c]; a1 += part[(size_t)(k + 64) * C + c]; a2 += part[(size_t)(k + 128) * C + c]; a3 += part[(size_t)(k + 192) * C + c];
    }
    for (; k < nparts; k += 64) a0 += part[(size_t)k * C + c];
  }
  sh[kl][cl] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (threadIdx.x < 16 && c < C) {
    float t = 0.f;
    for (int r = 0; r < 64; r++) t += sh[r][threadIdx.x];
    out[c] = accumulate ? out[c] + t : t;
  }
}

int dmvae_colsum_final(const float* part, float* out, int nparts, int C, int accumulate, hipStream_t stream) {  // used by groupnorm.hip
  if (nparts > 256) {
    hipLaunchKernelGGL(colsum_final_wide_kernel, dim3((C + 15) / 16), dim3(1024), 0, stream, part, out, nparts, C, accumulate);
    DMVAE_CHECK_LAUNCH();
    return 0;
  }
  hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 63) / 64), dim3(256), 0, stream, part, out, nparts, C, accumulate);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

// out[c] (+)= sum_r x[r][c] over a row-major bf16 matrix: the bias gradient on its own (callers that obtain the weight gradient with the operands'
// roles exchanged -- the sub-pixel form of Upsample's conv -- cannot take it from dmvae_conv2d_nhwc_wgrad).  workspace >= 512 * cols * 4 bytes.
extern "C" int dmvae_colsum_bf16(const void* x, void* out, void* workspace, size_t workspace_bytes, size_t rows, int cols, int accumulate,
                                 hipStream_t stream) {
  DMVAE_CHECK_ARG(x && out && workspace && rows > 0 && rows < (1ull << 31) && cols > 0 && cols % 8 == 0, "colsum_bf16: bad argument (cols must be a multiple of 8)");
  DMVAE_CHECK_ARG(workspace_bytes >= (size_t)512 * cols * sizeof(float), "colsum_bf16: workspace too small (need 512 * cols * 4 bytes)");
  return colsum_launch((const bf16*)x, (float*)out, (float*)workspace, (int)rows, cols, accumulate, stream);
}

// C[b][m][n] = alpha * sum_k A[b][k][m] * B[b][k][n]   (A: [K][M], B: [K][N] row-major bf16; C bf16 or f32)
extern "C" size_t dmvae_gemm_tn_batched_workspace(int M, int N, int K, int batch) {
  const int tiles = ((M + 127) / 128) * ((N + 127) / 128);
  int splits = pick_splits(K, tiles * batch);
  return (size_t)batch * splits * M * N * sizeof(float);
}
extern "C" int dmvae_gemm_tn_batched(const void* A, const void* B, void* C, void* workspace, size_t workspace_bytes, int M, int N,
                                     int K, int batch, long long a_bs, long long b_bs, long long c_bs, float alpha, int out_f32,
                                     hipStream_t stream) {
  DMVAE_CHECK_ARG(A && B && C && workspace, "gemm_tn_batched: null pointer");
  DMVAE_CHECK_ARG(M > 0 && M % 8 == 0 && N > 0 && N % 8 == 0 && K > 0 && batch > 0 && batch < 65536,
                  "gemm_tn_batched: need M%%8==0, N%%8==0 (M=%d N=%d K=%d batch=%d)", M, N, K, batch);
  DMVAE_CHECK_ARG(workspace_bytes >= dmvae_gemm_tn_batched_workspace(M, N, K, batch), "gemm_tn_batched: workspace too small");
  WgradArgs w;
  w.dy = (const bf16*)A; w.a = (const bf16*)B; w.slab = (float*)workspace;
  w.N = 1; w.Hi = 1; w.Wi = K; w.Ho = 1; w.Wo = K; w.Cin = N; w.Cout = M; w.ks = 1; w.so = 1; w.pd = 0; w.sd = 1; w.fl = 0; w.M = K;
  const int tiles = ((M + 127) / 128) * ((N + 127) / 128);
  int splits = pick_splits(K, tiles * batch);
  w.kchunk = (((K + splits - 1) / splits) + BKP - 1) / BKP * BKP;
  splits = (K + w.kchunk - 1) / w.kchunk;
  const size_t total = (size_t)M * N;
  w.dy_bs = a_bs; w.a_bs = b_bs; w.slab_bs = (long long)splits * total;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILEB);
    attr_done = true;
  }
  w.ntiles = tiles;
  hipLaunchKernelGGL(wgrad_kernel, dim3(splits * tiles, 1, batch), dim3(256), 4 * TILEB, stream, w);
  DMVAE_CHECK_LAUNCH();
  int rb = (int)((total + 255) / 256); if (rb > 1024) rb = 1024;
  if (out_f32)
    hipLaunchKernelGGL(tn_reduce_kernel<float>, dim3(rb, batch), dim3(256), 0, stream, w.slab, (float*)C, splits, total, w.slab_bs, c_bs, alpha);
  else
    hipLaunchKernelGGL(tn_reduce_kernel<bf16>, dim3(rb, batch), dim3(256), 0, stream, w.slab, (bf16*)C, splits, total, w.slab_bs, c_bs, alpha);
  DMVAE_CHECK_LAUNCH();
  return 0;
}
