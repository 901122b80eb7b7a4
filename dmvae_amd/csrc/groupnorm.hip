// GroupNorm(32 groups, eps) [+ swish] on NHWC bf16 activations: statistics, apply, and backward.
//
// Replaces nn.GroupNorm + swish at models/flux_ae.py:21-22,28,38,62,64,71-76,236,266-267 (the
// reference runs them as separate fp32 ATen kernels under autocast).  HBM-bound: coalesced 16-B
// loads (8 channels per lane), per-thread f32 accumulation, LDS block reduce, fixed-order final
// combine in f64 (deterministic; no atomics).
//
//   stats   : x -> (mean, rstd) per (image, group)                       reads 2 B/elem
//   apply   : a = act(x_hat*gamma+beta) -> bf16                           reads 2, writes 2 B/elem
//   bwd     : (da, x) -> A[n,c]=sum dy, B[n,c]=sum dy*x_hat   (reduce)    reads 4 B/elem
//             dx = rstd*(dy*gamma - (s1 + x_hat*s2)/m) [+ dres] (apply)   reads 4(+2), writes 2
//             with dy = da * swish'(x_hat*gamma+beta)
#include "common.h"
#include "dmvae_hip.h"

namespace dmvae_gn {

struct Geom {
  int HW, C, G, cpg, tp_shift, rows, ppc, nchunk;
};

__device__ __forceinline__ void load8(const bf16* p, float (&v)[8]) {
  const bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int e = 0; e < 8; e++) v[e] = (float)t[e];
}
// Raw 16-B load / later conversion, so that a loop can put the loads of TWO pixel rows in flight before it touches either: with one row per
// iteration and 5 (backward) to 8 waves per SIMD a CU holds 20-40 KB of reads in flight against the ~64 KB that 8 TB/s x ~2 us of latency asks of it.
// Loads of the streamed tensors are non-temporal: every pass here reads its operands once, 134-1074 MB of them, and without the hint they push the conv kernels'
// operands out of L2 / the Infinity Cache (same-box A/B of the whole step, three rounds: 67.18 / 66.99 / 67.00 -> 66.65 / 66.45 / 66.69 ms).  Stores stay plain
// (non-temporal stores measured +0.2 ms: the consumer conv wants the normalised tensor where plain stores leave it).
__device__ __forceinline__ bf16x8 ldraw(const bf16* p) {
  return __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p));
}
__device__ __forceinline__ void cvt8(const bf16x8& t, float (&v)[8]) {
#pragma unroll
  for (int e = 0; e < 8; e++) v[e] = (float)t[e];
}
__device__ __forceinline__ void store8(bf16* p, const float (&v)[8]) {
  bf16x8 t;
#pragma unroll
  for (int e = 0; e < 8; e++) t[e] = (bf16)v[e];
  *reinterpret_cast<bf16x8*>(p) = t;
}

// Reduce per-thread 2x8 channel accumulators over the block's pixel rows and write
// part[((n*nchunk+chunk)*C + c)*2 + {0,1}].
__device__ __forceinline__ void block_reduce_store(float (&s0)[8], float (&s1)[8], float* part, const Geom& g, int lane_c,
                                                   int prow, bool active) {
  __shared__ float red[256 * 16];
  const int tp = 1 << g.tp_shift;
  float* mine = red + (prow * tp + lane_c) * 16;
#pragma unroll
  for (int e = 0; e < 8; e++) { mine[e] = s0[e]; mine[8 + e] = s1[e]; }
  __syncthreads();
  // thread t < C handles channel t (may need two rounds when C > 256)
  for (int c = threadIdx.x; c < g.C; c += 256) {
    const int lc = c >> 3, e = c & 7;
    float a = 0.f, b = 0.f;
    for (int r = 0; r < g.rows; r++) {
      const float* q = red + (r * tp + lc) * 16;
      a += q[e]; b += q[8 + e];
    }
    float* o = part + ((size_t)(blockIdx.y * g.nchunk + blockIdx.x) * g.C + c) * 2;
    o[0] = a; o[1] = b;
  }
  (void)active;
}

__global__ __launch_bounds__(256) void stats_partial_kernel(const bf16* __restrict__ x, float* __restrict__ part, Geom g) {
  const int tp = 1 << g.tp_shift;
  const int lane_c = threadIdx.x & (tp - 1), prow = threadIdx.x >> g.tp_shift;
  const bool active = lane_c * 8 < g.C;
  const int n = blockIdx.y;
  const int p0 = blockIdx.x * g.ppc, p1 = min(p0 + g.ppc, g.HW);
  float s[8], ss[8];
#pragma unroll
  for (int e = 0; e < 8; e++) { s[e] = 0.f; ss[e] = 0.f; }
  if (active) {
    const bf16* base = x + (size_t)n * g.HW * g.C + lane_c * 8;
    auto acc = [&](const bf16x8& r) {
      float v[8];
      cvt8(r, v);
#pragma unroll
      for (int e = 0; e < 8; e++) { s[e] += v[e]; ss[e] += v[e] * v[e]; }
    };
    int p = p0 + prow;
    for (; p + 3 * g.rows < p1; p += 4 * g.rows) {       // four rows in flight; accumulated in the same order as one at a time
      const bf16x8 r0 = ldraw(base + (size_t)p * g.C), r1 = ldraw(base + (size_t)(p + g.rows) * g.C);
      const bf16x8 r2 = ldraw(base + (size_t)(p + 2 * g.rows) * g.C), r3 = ldraw(base + (size_t)(p + 3 * g.rows) * g.C);
      acc(r0); acc(r1); acc(r2); acc(r3);
    }
    for (; p < p1; p += g.rows) acc(ldraw(base + (size_t)p * g.C));
  }
  block_reduce_store(s, ss, part, g, lane_c, prow, active);
}

// one wave per (n, group): combine partials in f64
__global__ void stats_final_kernel(const float* __restrict__ part, float* __restrict__ stats, Geom g, int N, float eps) {
  const int wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wid >= N * g.G) return;
  const int n = wid / g.G, grp = wid % g.G;
  double s = 0.0, ss = 0.0;
  const int items = g.nchunk * g.cpg;
  for (int i = lane; i < items; i += 64) {
    const int ch = i / g.cpg, c = grp * g.cpg + i % g.cpg;
    const float* q = part + ((size_t)(n * g.nchunk + ch) * g.C + c) * 2;
    s += q[0]; ss += q[1];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); ss += __shfl_xor(ss, o, 64); }
  if (lane == 0) {
    const double cnt = (double)g.cpg * g.HW;
    const double mean = s / cnt;
    double var = ss / cnt - mean * mean;
    if (var < 0) var = 0;
    stats[wid * 2] = (float)mean;
    stats[wid * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

// one wave per (n, group): combine the per-tile, per-4-channel partials a conv epilogue left behind (conv_pp.hip, STATS) in f64
//   part[pixel tile][wave column 0..3][C / 4][2], tiles of `tp` pixels, whole tiles per image
__global__ void stats_from_quads_kernel(const float* __restrict__ part, float* __restrict__ stats, int N, int HW, int C, int G, int tp, float eps) {
  const int wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wid >= N * G) return;
  const int n = wid / G, grp = wid % G;
  const int qpg = C / G / 4, tpi = HW / tp, cq = C >> 2;   // quads per group, tiles per image
  double s = 0.0, ss = 0.0;
  const int items = tpi * 4 * qpg;
  for (int i = lane; i < items; i += 64) {
    const int q = i % qpg, r = i / qpg;                    // r = tile-in-image * 4 + wave column
    const float* p = part + ((size_t)(n * tpi * 4 + r) * cq + grp * qpg + q) * 2;
    s += p[0]; ss += p[1];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); ss += __shfl_xor(ss, o, 64); }
  if (lane == 0) {
    const double cnt = (double)(C / G) * HW;
    const double mean = s / cnt;
    double var = ss / cnt - mean * mean;
    if (var < 0) var = 0;
    stats[wid * 2] = (float)mean;
    stats[wid * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

template <int ACT>
__global__ __launch_bounds__(256) void apply_kernel(const bf16* __restrict__ x, const float* __restrict__ stats,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    bf16* __restrict__ y, Geom g) {
  const int tp = 1 << g.tp_shift;
  const int lane_c = threadIdx.x & (tp - 1), prow = threadIdx.x >> g.tp_shift;
  if (lane_c * 8 >= g.C) return;
  const int n = blockIdx.y;
  const int p0 = blockIdx.x * g.ppc, p1 = min(p0 + g.ppc, g.HW);
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; e++) {
    const int c = lane_c * 8 + e;
    const float* st = stats + ((size_t)n * g.G + c / g.cpg) * 2;
    sc[e] = st[1] * gamma[c];
    sh[e] = beta[c] - st[0] * sc[e];
  }
  const size_t base = (size_t)n * g.HW * g.C + lane_c * 8;
  auto row = [&](const bf16x8& r, int p) {
    float v[8];
    cvt8(r, v);
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const float t = v[e] * sc[e] + sh[e];
      v[e] = ACT == 1 ? t * sigmoidf_(t) : (ACT == 2 ? (t > 0.f ? t : 0.2f * t) : t);
    }
    store8(y + base + (size_t)p * g.C, v);
  };
  int p = p0 + prow;
  for (; p + g.rows < p1; p += 2 * g.rows) {
    const bf16x8 r0 = ldraw(x + base + (size_t)p * g.C), r1 = ldraw(x + base + (size_t)(p + g.rows) * g.C);
    row(r0, p); row(r1, p + g.rows);
  }
  if (p < p1) row(ldraw(x + base + (size_t)p * g.C), p);
}

template <int ACT>
__global__ __launch_bounds__(256) void bwd_partial_kernel(const bf16* __restrict__ da, const bf16* __restrict__ x,
                                                          const float* __restrict__ stats, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ part, Geom g) {
  const int tp = 1 << g.tp_shift;
  const int lane_c = threadIdx.x & (tp - 1), prow = threadIdx.x >> g.tp_shift;
  const bool active = lane_c * 8 < g.C;
  const int n = blockIdx.y;
  const int p0 = blockIdx.x * g.ppc, p1 = min(p0 + g.ppc, g.HW);
  float A[8], B[8];
#pragma unroll
  for (int e = 0; e < 8; e++) { A[e] = 0.f; B[e] = 0.f; }
  if (active) {
    float mu[8], rs[8], ga[8], be[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const int c = lane_c * 8 + e;
      const float* st = stats + ((size_t)n * g.G + c / g.cpg) * 2;
      mu[e] = st[0]; rs[e] = st[1]; ga[e] = gamma[c]; be[e] = beta[c];
    }
    const size_t base = (size_t)n * g.HW * g.C + lane_c * 8;
    auto row = [&](const bf16x8& rx, const bf16x8& rd) {
      float v[8], d[8];
      cvt8(rx, v);
      cvt8(rd, d);
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const float xh = (v[e] - mu[e]) * rs[e];
        float dy = d[e];
        if (ACT == 1) {
          const float t = xh * ga[e] + be[e];
          const float sg = sigmoidf_(t);
          dy *= sg * (1.f + t * (1.f - sg));
        } else if (ACT == 2) {
          dy = xh * ga[e] + be[e] > 0.f ? dy : 0.2f * dy;
        }
        A[e] += dy; B[e] += dy * xh;
      }
    };
    int p = p0 + prow;
    for (; p + g.rows < p1; p += 2 * g.rows) {          // two rows in flight; accumulated in the same order as one at a time
      const size_t o0 = base + (size_t)p * g.C, o1 = base + (size_t)(p + g.rows) * g.C;
      const bf16x8 x0 = ldraw(x + o0), d0 = ldraw(da + o0), x1 = ldraw(x + o1), d1 = ldraw(da + o1);
      row(x0, d0); row(x1, d1);
    }
    if (p < p1) row(ldraw(x + base + (size_t)p * g.C), ldraw(da + base + (size_t)p * g.C));
  }
  block_reduce_store(A, B, part, g, lane_c, prow, active);
}

// AB[n][c][2] = sum over chunks (f64 combine); then S[n][g][2] = sum_{c in g} gamma_c*(A,B)
__global__ void bwd_final_kernel(const float* __restrict__ part, const float* __restrict__ gamma, float* __restrict__ AB,
                                 float* __restrict__ S, Geom g, int N) {
  // one wave per (n, group)
  const int wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wid >= N * g.G) return;
  const int n = wid / g.G, grp = wid % g.G;
  double s1 = 0.0, s2 = 0.0;
  for (int ci = 0; ci < g.cpg; ci++) {
    const int c = grp * g.cpg + ci;
    double a = 0.0, b = 0.0;
    for (int ch = lane; ch < g.nchunk; ch += 64) {
      const float* q = part + ((size_t)(n * g.nchunk + ch) * g.C + c) * 2;
      a += q[0]; b += q[1];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
    if (lane == 0) { AB[((size_t)n * g.C + c) * 2] = (float)a; AB[((size_t)n * g.C + c) * 2 + 1] = (float)b; }
    s1 += (double)gamma[c] * a; s2 += (double)gamma[c] * b;
  }
  if (lane == 0) { S[wid * 2] = (float)s1; S[wid * 2 + 1] = (float)s2; }
}

// dgamma[c] (+)= sum_n B[n][c]; dbeta[c] (+)= sum_n A[n][c]
__global__ void bwd_param_kernel(const float* __restrict__ AB, float* __restrict__ dgamma, float* __restrict__ dbeta, int N, int C,
                                 int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double a = 0.0, b = 0.0;
  for (int n = 0; n < N; n++) { a += AB[((size_t)n * C + c) * 2]; b += AB[((size_t)n * C + c) * 2 + 1]; }
  dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)a;
  dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)b;
}

// COLS: the block also sums the (bf16-rounded) dx it stores per channel -> colpart[n * nchunk + chunk][C].  dx of a ResnetBlock's norm1 is the output gradient
// of the conv before it; when that conv is Upsample's in its sub-pixel form, its bias gradient (the column sums of dx) cannot come out of the weight-gradient
// kernel (operands' roles exchanged), and a separate pass over the full-resolution gradient costs 0.1-0.25 ms per layer.
template <int ACT, bool COLS = false>
__global__ __launch_bounds__(256) void bwd_apply_kernel(const bf16* __restrict__ da, const bf16* __restrict__ x,
                                                        const bf16* __restrict__ dres, const float* __restrict__ stats,
                                                        const float* __restrict__ S, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, bf16* __restrict__ dx, Geom g, float inv_count,
                                                        float* __restrict__ colpart = nullptr, const float* __restrict__ AB = nullptr,
                                                        float* __restrict__ dgamma = nullptr, float* __restrict__ dbeta = nullptr, int N = 0, int accumulate = 0) {
  if (AB && blockIdx.x == 0 && blockIdx.y == 0) {
    // the parameter gradients ride on this launch (bwd_param_kernel's arithmetic, one block of the thousands this grid has): one launch + one dispatch gap
    // less per GroupNorm backward, 30 of them per tokenizer step
    // (eight samples' loads in flight at a time, added in sample order: this block's extra time is on the launch's critical path when the whole grid is resident)
    for (int c = threadIdx.x; c < g.C; c += 256) {
      double a = 0.0, b = 0.0;
      int n = 0;
      for (; n + 8 <= N; n += 8) {
        f32x2 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = *reinterpret_cast<const f32x2*>(AB + ((size_t)(n + u) * g.C + c) * 2);
#pragma unroll
        for (int u = 0; u < 8; u++) { a += v[u][0]; b += v[u][1]; }
      }
      for (; n < N; n++) { a += AB[((size_t)n * g.C + c) * 2]; b += AB[((size_t)n * g.C + c) * 2 + 1]; }
      dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)a;
      dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)b;
    }
  }
  const int tp = 1 << g.tp_shift;
  const int lane_c = threadIdx.x & (tp - 1), prow = threadIdx.x >> g.tp_shift;
  if (!COLS && lane_c * 8 >= g.C) return;
  const bool live = lane_c * 8 < g.C;     // COLS: idle channel lanes stay for the block reduction
  const int n = blockIdx.y;
  const int p0 = blockIdx.x * g.ppc, p1 = live ? min(p0 + g.ppc, g.HW) : p0;
  const float inv_m = inv_count > 0.f ? inv_count : 1.0f / ((float)g.cpg * (float)g.HW);
  float mu[8], rs[8], ga[8], be[8], s1[8], s2[8];
  float cs[COLS ? 8 : 1];
  if constexpr (COLS) {
#pragma unroll
    for (int e = 0; e < 8; e++) cs[e] = 0.f;
  }
#pragma unroll
  for (int e = 0; e < 8; e++) {
    const int c = live ? lane_c * 8 + e : 0;
    const size_t gi = ((size_t)n * g.G + c / g.cpg) * 2;
    mu[e] = stats[gi]; rs[e] = stats[gi + 1]; ga[e] = gamma[c]; be[e] = beta[c];
    s1[e] = S[gi] * inv_m; s2[e] = S[gi + 1] * inv_m;
  }
  const size_t base = (size_t)n * g.HW * g.C + lane_c * 8;
  auto row = [&](const bf16x8& rx, const bf16x8& rd, const bf16x8& ro, int p) {
    float v[8], d[8], o[8];
    cvt8(rx, v);
    cvt8(rd, d);
    if (dres) cvt8(ro, o);
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const float xh = (v[e] - mu[e]) * rs[e];
      float dy = d[e];
      if (ACT == 1) {
        const float t = xh * ga[e] + be[e];
        const float sg = sigmoidf_(t);
        dy *= sg * (1.f + t * (1.f - sg));
      } else if (ACT == 2) {
        dy = xh * ga[e] + be[e] > 0.f ? dy : 0.2f * dy;
      }
      const float r = rs[e] * (dy * ga[e] - s1[e] - xh * s2[e]);
      o[e] = dres ? o[e] + r : r;
      if constexpr (COLS) cs[e] += (float)(bf16)o[e];      // what is stored, as a later pass over dx would read it
    }
    store8(dx + base + (size_t)p * g.C, o);
  };
  int p = p0 + prow;
  for (; p + g.rows < p1; p += 2 * g.rows) {
    const size_t o0 = base + (size_t)p * g.C, o1 = base + (size_t)(p + g.rows) * g.C;
    const bf16x8 x0 = ldraw(x + o0), d0 = ldraw(da + o0), x1 = ldraw(x + o1), d1 = ldraw(da + o1);
    bf16x8 q0 = x0, q1 = x1;
    if (dres) { q0 = ldraw(dres + o0); q1 = ldraw(dres + o1); }
    row(x0, d0, q0, p); row(x1, d1, q1, p + g.rows);
  }
  if (p < p1) {
    const size_t o0 = base + (size_t)p * g.C;
    const bf16x8 x0 = ldraw(x + o0), d0 = ldraw(da + o0);
    row(x0, d0, dres ? ldraw(dres + o0) : x0, p);
  }
  if constexpr (COLS) {  // fold the block's pixel rows in fixed order: [rows][tp * 8] through LDS
    __shared__ float red[256 * 8];
#pragma unroll
    for (int e = 0; e < 8; e++) red[(prow * tp + lane_c) * 8 + e] = cs[e];
    __syncthreads();
    for (int c = threadIdx.x; c < tp * 8; c += 256) {
      float a = 0.f;
      for (int r = 0; r < g.rows; r++) a += red[(r * tp + (c >> 3)) * 8 + (c & 7)];
      if (c < g.C) colpart[((size_t)n * g.nchunk + blockIdx.x) * g.C + c] = a;
    }
  }
}

static int make_geom(Geom& g, int N, int HW, int C, int G) {
  if (C <= 0 || C % 8 != 0 || C > 512 || G <= 0 || C % G != 0 || HW <= 0 || N <= 0) return -1;
  g.HW = HW; g.C = C; g.G = G; g.cpg = C / G;
  int tp = 1, sh = 0;
  while (tp < C / 8) { tp <<= 1; sh++; }
  g.tp_shift = sh; g.rows = 256 / tp;
  // chunks per image: target ~2048 blocks overall, >= 4 row-iterations per block
  int nchunk = (2048 + N - 1) / N;
  const int maxc = (HW + g.rows * 4 - 1) / (g.rows * 4);
  if (nchunk > maxc) nchunk = maxc;
  if (nchunk > 2048) nchunk = 2048;   // N >= 32 (every decoder shape): <= 64 chunks per image; a single "image" (BatchNorm use) still fills the chip
  if (nchunk < 1) nchunk = 1;
  g.ppc = (HW + nchunk - 1) / nchunk;
  g.nchunk = (HW + g.ppc - 1) / g.ppc;
  return 0;
}

}  // namespace dmvae_gn
using namespace dmvae_gn;

extern "C" size_t dmvae_groupnorm_workspace(int n, int hw, int c, int groups) {
  Geom g;
  if (make_geom(g, n, hw, c, groups)) return 0;
  // partials + AB + S
  return ((size_t)n * g.nchunk * c * 2 + (size_t)n * c * 2 + (size_t)n * groups * 2) * sizeof(float);
}

extern "C" int dmvae_groupnorm_stats(const void* x, void* stats, void* workspace, size_t workspace_bytes, int n, int hw, int c,
                                     int groups, float eps, hipStream_t stream) {
  Geom g;
  DMVAE_CHECK_ARG(x && stats && workspace, "groupnorm_stats: null pointer");
  DMVAE_CHECK_ARG(make_geom(g, n, hw, c, groups) == 0, "groupnorm_stats: unsupported shape n=%d hw=%d c=%d groups=%d", n, hw, c, groups);
  DMVAE_CHECK_ARG(workspace_bytes >= dmvae_groupnorm_workspace(n, hw, c, groups), "groupnorm_stats: workspace too small");
  hipLaunchKernelGGL(stats_partial_kernel, dim3(g.nchunk, n), dim3(256), 0, stream, (const bf16*)x, (float*)workspace, g);
  DMVAE_CHECK_LAUNCH();
  const int waves = n * groups;
  hipLaunchKernelGGL(stats_final_kernel, dim3((waves + 3) / 4), dim3(256), 0, stream, (const float*)workspace, (float*)stats, g, n, eps);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

int dmvae_gn_stats_from_quads(const float* part, float* stats, int n, int hw, int c, int groups, int tp, float eps, hipStream_t stream) {
  DMVAE_CHECK_ARG(part && stats && groups > 0 && c % groups == 0 && (c / groups) % 4 == 0 && tp > 0 && hw % tp == 0, "gn_stats_from_quads: bad shape");
  const int waves = n * groups;
  hipLaunchKernelGGL(stats_from_quads_kernel, dim3((waves + 3) / 4), dim3(256), 0, stream, part, stats, n, hw, c, groups, tp, eps);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

extern "C" int dmvae_groupnorm_apply(const void* x, const void* stats, const void* gamma, const void* beta, void* y, int n, int hw,
                                     int c, int groups, int act, hipStream_t stream) {
  Geom g;
  DMVAE_CHECK_ARG(x && stats && gamma && beta && y, "groupnorm_apply: null pointer");
  DMVAE_CHECK_ARG(act >= 0 && act <= 2, "groupnorm_apply: act must be 0 (none), 1 (swish) or 2 (LeakyReLU 0.2)");
  DMVAE_CHECK_ARG(make_geom(g, n, hw, c, groups) == 0, "groupnorm_apply: unsupported shape n=%d hw=%d c=%d groups=%d", n, hw, c, groups);
  const dim3 grid(g.nchunk, n);
#define DMVAE_GN_APPLY(A) hipLaunchKernelGGL(apply_kernel<A>, grid, dim3(256), 0, stream, (const bf16*)x, (const float*)stats, \
                                             (const float*)gamma, (const float*)beta, (bf16*)y, g)
  if (act == 1) DMVAE_GN_APPLY(1); else if (act == 2) DMVAE_GN_APPLY(2); else DMVAE_GN_APPLY(0);
#undef DMVAE_GN_APPLY
  DMVAE_CHECK_LAUNCH();
  return 0;
}

// Reduction half of the backward: sums[n][groups][2] = (sum g, sum g*x_hat) with g = da*act'(.)*gamma, plus dgamma / dbeta.
static int bwd_reduce_impl(const void* da, const void* x, const void* stats, const void* gamma, const void* beta, void* sums, void* dgamma, void* dbeta,
                           void* workspace, size_t workspace_bytes, int n, int hw, int c, int groups, int act, int accumulate, bool with_param, hipStream_t stream);
extern "C" int dmvae_groupnorm_bwd_reduce(const void* da, const void* x, const void* stats, const void* gamma, const void* beta,
                                          void* sums, void* dgamma, void* dbeta, void* workspace, size_t workspace_bytes, int n, int hw,
                                          int c, int groups, int act, int accumulate, hipStream_t stream) {
  return bwd_reduce_impl(da, x, stats, gamma, beta, sums, dgamma, dbeta, workspace, workspace_bytes, n, hw, c, groups, act, accumulate, true, stream);
}
// with_param = false: the caller's apply launch computes dgamma / dbeta from AB (bwd_apply_kernel's prologue)
static int bwd_reduce_impl(const void* da, const void* x, const void* stats, const void* gamma, const void* beta, void* sums, void* dgamma, void* dbeta,
                           void* workspace, size_t workspace_bytes, int n, int hw, int c, int groups, int act, int accumulate, bool with_param, hipStream_t stream) {
  Geom g;
  DMVAE_CHECK_ARG(da && x && stats && gamma && beta && sums && workspace, "groupnorm_bwd_reduce: null pointer");
  DMVAE_CHECK_ARG(act >= 0 && act <= 2, "groupnorm_bwd_reduce: act must be 0, 1 or 2");
  DMVAE_CHECK_ARG(make_geom(g, n, hw, c, groups) == 0, "groupnorm_bwd_reduce: unsupported shape n=%d hw=%d c=%d groups=%d", n, hw, c, groups);
  DMVAE_CHECK_ARG(workspace_bytes >= dmvae_groupnorm_workspace(n, hw, c, groups), "groupnorm_bwd_reduce: workspace too small");
  float* part = (float*)workspace;
  float* AB = part + (size_t)n * g.nchunk * c * 2;
  const dim3 grid(g.nchunk, n);
#define DMVAE_GN_PART(A) hipLaunchKernelGGL(bwd_partial_kernel<A>, grid, dim3(256), 0, stream, (const bf16*)da, (const bf16*)x, \
                                            (const float*)stats, (const float*)gamma, (const float*)beta, part, g)
  if (act == 1) DMVAE_GN_PART(1); else if (act == 2) DMVAE_GN_PART(2); else DMVAE_GN_PART(0);
#undef DMVAE_GN_PART
  DMVAE_CHECK_LAUNCH();
  const int waves = n * groups;
  hipLaunchKernelGGL(bwd_final_kernel, dim3((waves + 3) / 4), dim3(256), 0, stream, part, (const float*)gamma, AB, (float*)sums, g, n);
  DMVAE_CHECK_LAUNCH();
  if (dgamma && dbeta && with_param) {
    hipLaunchKernelGGL(bwd_param_kernel, dim3((c + 255) / 256), dim3(256), 0, stream, AB, (float*)dgamma, (float*)dbeta, n, c, accumulate);
    DMVAE_CHECK_LAUNCH();
  }
  return 0;
}

// Elementwise half: dx = rstd*(g - (s1 + x_hat*s2)*inv_count) [+ dres]; inv_count <= 0 selects 1/(hw*c/groups).
extern "C" int dmvae_groupnorm_bwd_apply(const void* da, const void* x, const void* dres, const void* stats, const void* sums,
                                         const void* gamma, const void* beta, void* dx, int n, int hw, int c, int groups, int act,
                                         float inv_count, hipStream_t stream) {
  Geom g;
  DMVAE_CHECK_ARG(da && x && stats && sums && gamma && beta && dx, "groupnorm_bwd_apply: null pointer");
  DMVAE_CHECK_ARG(act >= 0 && act <= 2, "groupnorm_bwd_apply: act must be 0, 1 or 2");
  DMVAE_CHECK_ARG(make_geom(g, n, hw, c, groups) == 0, "groupnorm_bwd_apply: unsupported shape n=%d hw=%d c=%d groups=%d", n, hw, c, groups);
  const dim3 grid(g.nchunk, n);
#define DMVAE_GN_BAPPLY(A) hipLaunchKernelGGL(bwd_apply_kernel<A>, grid, dim3(256), 0, stream, (const bf16*)da, (const bf16*)x, (const bf16*)dres, \
                                              (const float*)stats, (const float*)sums, (const float*)gamma, (const float*)beta, (bf16*)dx, g, inv_count)
  if (act == 1) DMVAE_GN_BAPPLY(1); else if (act == 2) DMVAE_GN_BAPPLY(2); else DMVAE_GN_BAPPLY(0);
#undef DMVAE_GN_BAPPLY
  DMVAE_CHECK_LAUNCH();
  return 0;
}

extern "C" int dmvae_groupnorm_bwd(const void* da, const void* x, const void* dres, const void* stats, const void* gamma,
                                   const void* beta, void* dx, void* dgamma, void* dbeta, void* workspace, size_t workspace_bytes,
                                   int n, int hw, int c, int groups, int act, int accumulate, hipStream_t stream) {
  Geom g;
  DMVAE_CHECK_ARG(da && x && stats && gamma && beta && dx && workspace, "groupnorm_bwd: null pointer");
  DMVAE_CHECK_ARG(act >= 0 && act <= 2, "groupnorm_bwd: act must be 0, 1 or 2");
  DMVAE_CHECK_ARG(make_geom(g, n, hw, c, groups) == 0, "groupnorm_bwd: unsupported shape n=%d hw=%d c=%d groups=%d", n, hw, c, groups);
  DMVAE_CHECK_ARG(workspace_bytes >= dmvae_groupnorm_workspace(n, hw, c, groups), "groupnorm_bwd: workspace too small");
  float* AB = (float*)workspace + (size_t)n * g.nchunk * c * 2;
  float* S = AB + (size_t)n * c * 2;
  int rc = bwd_reduce_impl(da, x, stats, gamma, beta, S, dgamma, dbeta, workspace, workspace_bytes, n, hw, c, groups, act, accumulate, false, stream);
  if (rc) return rc;
  const bool par = dgamma && dbeta;
  const dim3 grid(g.nchunk, n);
#define DMVAE_GN_BAPPLYP(A) hipLaunchKernelGGL((bwd_apply_kernel<A, false>), grid, dim3(256), 0, stream, (const bf16*)da, (const bf16*)x, (const bf16*)dres, \
                                               (const float*)stats, (const float*)S, (const float*)gamma, (const float*)beta, (bf16*)dx, g, 0.f, (float*)nullptr, \
                                               par ? (const float*)AB : (const float*)nullptr, (float*)dgamma, (float*)dbeta, n, accumulate)
  if (act == 1) DMVAE_GN_BAPPLYP(1); else if (act == 2) DMVAE_GN_BAPPLYP(2); else DMVAE_GN_BAPPLYP(0);
#undef DMVAE_GN_BAPPLYP
  DMVAE_CHECK_LAUNCH();
  return 0;
}

int dmvae_colsum_final(const float* part, float* out, int nparts, int C, int accumulate, hipStream_t stream);  // conv_wgrad.hip

// dmvae_groupnorm_bwd that also returns colsum[c] = sum over (n, hw) of the dx it stores: the bias gradient of the conv whose output gradient dx is.
extern "C" int dmvae_groupnorm_bwd_colsum(const void* da, const void* x, const void* dres, const void* stats, const void* gamma, const void* beta, void* dx,
                                          void* dgamma, void* dbeta, void* colsum, void* workspace, size_t workspace_bytes, int n, int hw, int c, int groups,
                                          int act, int accumulate, int colsum_accumulate, hipStream_t stream) {
  Geom g;
  DMVAE_CHECK_ARG(da && x && stats && gamma && beta && dx && colsum && workspace, "groupnorm_bwd_colsum: null pointer");
  DMVAE_CHECK_ARG(act >= 0 && act <= 2, "groupnorm_bwd_colsum: act must be 0, 1 or 2");
  DMVAE_CHECK_ARG(make_geom(g, n, hw, c, groups) == 0, "groupnorm_bwd_colsum: unsupported shape n=%d hw=%d c=%d groups=%d", n, hw, c, groups);
  DMVAE_CHECK_ARG(workspace_bytes >= dmvae_groupnorm_workspace(n, hw, c, groups), "groupnorm_bwd_colsum: workspace too small");
  float* AB = (float*)workspace + (size_t)n * g.nchunk * c * 2;
  float* S = AB + (size_t)n * c * 2;
  int rc = bwd_reduce_impl(da, x, stats, gamma, beta, S, dgamma, dbeta, workspace, workspace_bytes, n, hw, c, groups, act, accumulate, false, stream);
  if (rc) return rc;
  const bool par = dgamma && dbeta;
  float* colpart = (float*)workspace;   // the reduce half's partials are consumed by now (stream order): their space takes the column partials (AB lies behind them)
  const dim3 grid(g.nchunk, n);
#define DMVAE_GN_BAPPLYC(A) hipLaunchKernelGGL((bwd_apply_kernel<A, true>), grid, dim3(256), 0, stream, (const bf16*)da, (const bf16*)x, (const bf16*)dres, \
                                               (const float*)stats, (const float*)S, (const float*)gamma, (const float*)beta, (bf16*)dx, g, 0.f, colpart, \
                                               par ? (const float*)AB : (const float*)nullptr, (float*)dgamma, (float*)dbeta, n, accumulate)
  if (act == 1) DMVAE_GN_BAPPLYC(1); else if (act == 2) DMVAE_GN_BAPPLYC(2); else DMVAE_GN_BAPPLYC(0);
#undef DMVAE_GN_BAPPLYC
  DMVAE_CHECK_LAUNCH();
  return dmvae_colsum_final(colpart, (float*)colsum, n * g.nchunk, c, colsum_accumulate, stream);
}
