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
  // eight loads in flight per lane, added in the order of one at a time (a lane walks 8 items at 128 channels @ 256 x 256: eight dependent L2 round trips
  // made this launch 6-16 us, 33 of them per step); items past the end add +0.0
  for (int i0 = lane; i0 < items; i0 += 64 * 8) {
    f32x2 v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int i = i0 + 64 * u;
      const int q = i % qpg, r = i / qpg;                  // r = tile-in-image * 4 + wave column
      v[u] = i < items ? *reinterpret_cast<const f32x2*>(part + ((size_t)(n * tpi * 4 + r) * cq + grp * qpg + q) * 2) : f32x2{0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < 8; u++) { s += v[u][0]; ss += v[u][1]; }
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
  // Eight channels at a time: 16 per-lane f64 partial sums (over the lane's chunks, in order) are turned through LDS -- value v of lane l at [v][l], rows padded to
  // 65 doubles -- and lane 16 q + v adds the 16 lanes of quarter q in lane order; two shuffles add the quarters.  (The first form ran a six-step f64 butterfly on
  // all 16 values: ~ 200 cross-lane moves per batch through the LDS crossbar, 12 us per launch at 512 channels.)  Fixed order: deterministic.
  __shared__ double red[4][16][65];
  double (*rw)[65] = red[threadIdx.x >> 6];
  for (int c0 = 0; c0 < g.cpg; c0 += 8) {
    double a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; u++) { a[u] = 0.0; b[u] = 0.0; }
    for (int ch = lane; ch < g.nchunk; ch += 64) {
      f32x2 v[8];
#pragma unroll
      for (int u = 0; u < 8; u++)
        v[u] = c0 + u < g.cpg ? *reinterpret_cast<const f32x2*>(part + ((size_t)(n * g.nchunk + ch) * g.C + grp * g.cpg + c0 + u) * 2) : f32x2{0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 8; u++) { a[u] += v[u][0]; b[u] += v[u][1]; }
    }
#pragma unroll
    for (int u = 0; u < 8; u++) { rw[2 * u][lane] = a[u]; rw[2 * u + 1][lane] = b[u]; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int v = lane & 15, q = lane >> 4;
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < 16; i++) t += rw[v][16 * q + i];
    t += __shfl_xor(t, 16, 64);
    t += __shfl_xor(t, 32, 64);                            // every lane: the total of value (lane & 15) = (channel c0 + (v >> 1), sum A / B)
    {
      const int u = v >> 1, c = grp * g.cpg + c0 + u;
      if (lane < 16 && c0 + u < g.cpg) AB[((size_t)n * g.C + c) * 2 + (v & 1)] = (float)t;
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const double au = __shfl(t, 2 * u, 64), bu = __shfl(t, 2 * u + 1, 64);
      if (c0 + u < g.cpg) {
        const int c = grp * g.cpg + c0 + u;
        s1 += (double)gamma[c] * au; s2 += (double)gamma[c] * bu;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the rows are rewritten by the next batch
    __builtin_amdgcn_wave_barrier();
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

// ---- conv_out(swish(norm_out(x))) backward with respect to x, gamma, beta (flux_ae.py:266-268) -----------------------------------------------------------
// The decoder's last GroupNorm sits on its largest tensor ([N, 256, 256, 128]: 537 MB at B = 32) and its output gradient `da` is the input gradient of a conv with
// THREE output channels: da[q][ci] = sum over (tap, co) of dy[q + off(tap)][co] * W[co][ci][tap] -- 27 multiply-adds per element from 54 bytes of dy per pixel.
// Stored, `da` is written once (by an input-gradient conv whose reduction dimension is 89 % zero padding: 203 us) and read twice (by the two GroupNorm backward
// passes): 1.6 GB.  Here both passes evaluate it in place on the matrix cores from a zero-bordered 4-channel bf16 copy of dy ([N][H + 2][W + 2][4], 17 MB,
// cache-resident: convout_pad_kernel; the same rounding of dy as the stored-operand route) and round the result to bf16 where the stored tensor would have been:
// the passes read x only.
//   A wave owns 16-pixel runs of an image row.  Per run: 16 x (taps 0-7 x 4 channels) as the B operand of eight v_mfma_f32_16x16x32_bf16 (one per 16 channels;
// lane (p, kg) supplies the 8-B pixels of taps 2 kg and 2 kg + 1 of pixel p: unconditional loads, the border is in the copy) plus eight v_mfma_f32_16x16x16_bf16
// for tap 8; the weights (A operands) sit in LDS as the lanes read them (12 KB per block, converted from the f32 parameter at block start).  D comes out as
// lane (p, kg) <-> pixel p, channels 16 f + 4 kg + i; it goes through a 16 x 128 bf16 tile in LDS (per wave, rows padded to 272 B) and comes back in the
// elementwise kernels' layout -- lane <-> pixel 4 r + (lane >> 4), channels 8 (lane & 15) + 0..7 -- in which x is loaded and dx is stored as 1-KB contiguous
// wave accesses, and from there on the arithmetic IS bwd_partial_kernel's / bwd_apply_kernel's.  (First version: lane <-> the MFMA's own layout, 64-B pieces of
// 16 rows per access, a block's four waves on four channel quarters of the same pixels: 217 + 310 us, no faster with the MFMA, the sigmoid and the dy loads
// all compiled out -- the access pattern was the cost.)
__global__ void convout_pad_kernel(const float* __restrict__ dy, bf16* __restrict__ out, int N, int H, int W) {
  const size_t total = (size_t)N * (H + 2) * (W + 2);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int xx = (int)(i % (W + 2)) - 1;
    const size_t r = i / (W + 2);
    const int yy = (int)(r % (H + 2)) - 1, n = (int)(r / (H + 2));
    bf16x4 v = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
    if (xx >= 0 && xx < W && yy >= 0 && yy < H) {
      const float* s = dy + ((size_t)n * 3 * H + yy) * W + xx;
      v[0] = (bf16)s[0]; v[1] = (bf16)s[(size_t)H * W]; v[2] = (bf16)s[(size_t)2 * H * W];
    }
    *reinterpret_cast<bf16x4*>(out + i * 4) = v;
  }
}

constexpr int CO_ROW = 272;                               // bytes per pixel row of a wave's tile: 256 + 16 (b64 writes of 16 rows: 2-way conflicts at most)
constexpr int CO_W1 = 8 * 64 * 16, CO_W2 = 8 * 64 * 8;    // A operands of the two steps, [fragment][lane]
constexpr int CO_LDS = CO_W1 + CO_W2 + 4 * 16 * CO_ROW;   // 29696 B; the partial pass's block reduction (16 KB) reuses it

template <bool APPLY>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void convout_bwd_kernel(const bf16* __restrict__ dyp, const float* __restrict__ wt, const bf16* __restrict__ x,
                                                          const float* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ part, const float* __restrict__ S, bf16* __restrict__ dx, Geom g, int H, int W,
                                                          const float* __restrict__ AB, float* __restrict__ dgamma, float* __restrict__ dbeta, int N, int accumulate) {
  __shared__ __attribute__((aligned(16))) char smem[CO_LDS];
  if (APPLY && AB && blockIdx.x == 0 && blockIdx.y == 0) {   // the parameter gradients ride on this launch, as in bwd_apply_kernel
    for (int c = threadIdx.x; c < g.C; c += 256) {
      double a = 0.0, b = 0.0;
      for (int n = 0; n < N; n++) { a += AB[((size_t)n * g.C + c) * 2]; b += AB[((size_t)n * g.C + c) * 2 + 1]; }
      dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)a;
      dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)b;
    }
  }
  const int wv = threadIdx.x >> 6, l = threadIdx.x & 63, p = l & 15, kg = l >> 4;
  const int n = blockIdx.y, C = g.C;
  // A operands: row r = lane & 15 of fragment f is channel 16 f + r; step 1: k = 8 kg + kk <-> tap 2 kg + (kk >> 2), output channel kk & 3; step 2 (K = 16):
  // k = 4 kg + kk <-> tap 8 + kg (tap 8 only), output channel kk
  for (int i = threadIdx.x; i < 8 * 64; i += 256) {
    const int f = i >> 6, ll = i & 63, ch = 16 * f + (ll & 15), k8 = ll >> 4;
    bf16x8 a;
    bf16x4 a2;
#pragma unroll
    for (int kk = 0; kk < 8; kk++) {
      const int co = kk & 3, tap = 2 * k8 + (kk >> 2);
      a[kk] = (bf16)(co < 3 ? wt[((size_t)co * C + ch) * 9 + tap] : 0.f);
    }
#pragma unroll
    for (int kk = 0; kk < 4; kk++) a2[kk] = (bf16)((kk < 3 && k8 == 0) ? wt[((size_t)kk * C + ch) * 9 + 8] : 0.f);
    *reinterpret_cast<bf16x8*>(smem + i * 16) = a;
    *reinterpret_cast<bf16x4*>(smem + CO_W1 + i * 8) = a2;
  }
  __syncthreads();
  char* tile = smem + CO_W1 + CO_W2 + wv * 16 * CO_ROW;
  const int cb = 8 * p;                                   // elementwise layout: this lane's eight channels (and pixel 4 r + kg of a run)
  // per-group values once per channel PAIR (cpg is even: a pair never straddles two groups)
  float mu[4], rs[4], s1[APPLY ? 4 : 1], s2[APPLY ? 4 : 1];
  f32x2 ga[4], be[4];
  const float inv_m = 1.0f / ((float)g.cpg * (float)g.HW);
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const int c = cb + 2 * e;
    const size_t gi = ((size_t)n * g.G + c / g.cpg) * 2;
    mu[e] = stats[gi]; rs[e] = stats[gi + 1];
    ga[e] = f32x2{gamma[c], gamma[c + 1]}; be[e] = f32x2{beta[c], beta[c + 1]};
    if constexpr (APPLY) { s1[e] = S[gi] * inv_m; s2[e] = S[gi + 1] * inv_m; }
  }
  f32x2 A[4], B[4];
#pragma unroll
  for (int e = 0; e < 4; e++) { A[e] = f32x2{0.f, 0.f}; B[e] = f32x2{0.f, 0.f}; }
  // tap t = (ky, kx) = (t / 3, t % 3) reads dy at (y + 1 - ky, x + 1 - kx): element offsets inside the bordered copy, relative to the pixel itself
  const int Wp = W + 2;
  const int t0 = 2 * kg, t1 = 2 * kg + 1;
  const int lo0 = ((1 - t0 / 3) * Wp + (1 - t0 % 3)) * 4, lo1 = ((1 - t1 / 3) * Wp + (1 - t1 % 3)) * 4, lo8 = (-Wp - 1) * 4;
  const bf16* dn = dyp + ((size_t)n * (H + 2) * Wp + Wp + 1 + p) * 4;      // pixel (0, p) of image n
  const bf16* xn = x + ((size_t)n * g.HW + kg) * C + cb;
  bf16* dxn = APPLY ? dx + ((size_t)n * g.HW + kg) * C + cb : nullptr;
  const int p0 = blockIdx.x * g.ppc, p1 = min(p0 + g.ppc, g.HW);     // ppc % 16 == 0, HW % 16 == 0, W % 16 == 0: whole runs inside one image row
  for (int q = p0 + 16 * wv; q < p1; q += 64) {
    bf16x8 xr[4];
#pragma unroll
    for (int r = 0; r < 4; r++) xr[r] = ldraw(xn + (size_t)(q + 4 * r) * C);
    const int y = q / W, x0 = q - y * W;
    const bf16* d = dn + (y * Wp + x0) * 4;
    const bf16x4 d0 = *reinterpret_cast<const bf16x4*>(d + lo0), d1 = *reinterpret_cast<const bf16x4*>(d + lo1);
    const bf16x4 d8 = *reinterpret_cast<const bf16x4*>(d + lo8);      // every lane loads it; only k group 0's weights are non-zero there
    // The sixteen MFMAs of a run are issued as inline assembly, four fragments at a time: four independent 16x16x32 products (C = 0) into early-clobber
    // results, then the four tap-8 products (16x16x16) accumulated in place.  Through the builtins, fragment by fragment, hipcc emitted
    //     v_mfma_f32_16x16x32_bf16 D, A, B, 0 ; v_mfma_f32_16x16x16_bf16 D, A2, B2, D
    // back to back: an accumulate chain through two DIFFERENT opcodes gets no wait states from this toolchain and no interlock from the hardware -- the second
    // read C before the first (8 passes) had written it and the first product was lost (tools/check_mfma_chain.py audits every kernel's assembly for the pair).
    // In this order every dependent instruction is three MFMAs (>= 96 cycles) behind its producer; the trailing s_nop covers the last one before the VALU reads it.
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    typedef int i32x2 __attribute__((ext_vector_type(2)));
    union { bf16x8 v; i32x4 r; } b0;
    union { bf16x4 v; i32x2 r; } b1;
    b0.v = __builtin_shufflevector(d0, d1, 0, 1, 2, 3, 4, 5, 6, 7);
    b1.v = d8;
#pragma unroll
    for (int h = 0; h < 2; h++) {
      f32x4 acc[4];
#pragma unroll
      for (int f = 0; f < 4; f++) {
        union { bf16x8 v; i32x4 r; } a;
        a.v = *reinterpret_cast<const bf16x8*>(smem + ((4 * h + f) * 64 + l) * 16);
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(acc[f]) : "v"(a.r), "v"(b0.r));
      }
#pragma unroll
      for (int f = 0; f < 4; f++) {
        union { bf16x4 v; i32x2 r; } a2;
        a2.v = *reinterpret_cast<const bf16x4*>(smem + CO_W1 + ((4 * h + f) * 64 + l) * 8);
        asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0" : "+v"(acc[f]) : "v"(a2.r), "v"(b1.r));
      }
      asm volatile("s_nop 15" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
#pragma unroll
      for (int f = 0; f < 4; f++) {
        const bf16x4 o = {(bf16)acc[f][0], (bf16)acc[f][1], (bf16)acc[f][2], (bf16)acc[f][3]};      // the stored tensor's rounding site
        *reinterpret_cast<bf16x4*>(tile + p * CO_ROW + (16 * (4 * h + f) + 4 * kg) * 2) = o;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const bf16x8 dr = *reinterpret_cast<const bf16x8*>(tile + (4 * r + kg) * CO_ROW + cb * 2);
      bf16x8 ob;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const f32x2 v = {(float)xr[r][2 * e], (float)xr[r][2 * e + 1]};
        const f32x2 dd = {(float)dr[2 * e], (float)dr[2 * e + 1]};
        const f32x2 xh = (v - f32x2{mu[e], mu[e]}) * f32x2{rs[e], rs[e]};
        const f32x2 t = xh * ga[e] + be[e];
        const f32x2 sg = {sigmoidf_(t[0]), sigmoidf_(t[1])};
        const f32x2 dy = dd * (sg * (1.f + t * (1.f - sg)));
        if constexpr (APPLY) {
          const f32x2 o = f32x2{rs[e], rs[e]} * (dy * ga[e] - f32x2{s1[e], s1[e]} - xh * f32x2{s2[e], s2[e]});
          ob[2 * e] = (bf16)o[0]; ob[2 * e + 1] = (bf16)o[1];
        } else {
          A[e] += dy; B[e] += dy * xh;
        }
      }
      if constexpr (APPLY) *reinterpret_cast<bf16x8*>(dxn + (size_t)(q + 4 * r) * C) = ob;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the tile is rewritten by the next run
    __builtin_amdgcn_wave_barrier();
  }
  if constexpr (!APPLY) {   // block_reduce_store's fold in this kernel's LDS: [thread][16] -> per channel over the 16 threads that hold it, in thread order
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int e = 0; e < 8; e++) { red[threadIdx.x * 16 + e] = A[e >> 1][e & 1]; red[threadIdx.x * 16 + 8 + e] = B[e >> 1][e & 1]; }
    __syncthreads();
    if (threadIdx.x < 128) {
      const int c = threadIdx.x, lc = c >> 3, e = c & 7;
      float a = 0.f, b = 0.f;
      for (int r = 0; r < 16; r++) { const float* qq = red + ((r * 16 + lc) * 16); a += qq[e]; b += qq[8 + e]; }
      float* o = part + ((size_t)(n * g.nchunk + blockIdx.x) * C + c) * 2;
      o[0] = a; o[1] = b;
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

// The reduction half for an apply launch that lives in another file (norm_short.hip): partials -> AB [n][c][2] and S [n][groups][2] inside `workspace`
// (dmvae_groupnorm_workspace bytes), no parameter-gradient launch (the caller's apply launch adds AB up).
int dmvae_gn_bwd_reduce_parts(const void* da, const void* x, const void* stats, const void* gamma, const void* beta, void* workspace, size_t workspace_bytes,
                              int n, int hw, int c, int groups, int act, float** AB, float** S, hipStream_t stream) {
  Geom g;
  DMVAE_CHECK_ARG(make_geom(g, n, hw, c, groups) == 0, "groupnorm_bwd_reduce: unsupported shape n=%d hw=%d c=%d groups=%d", n, hw, c, groups);
  *AB = (float*)workspace + (size_t)n * g.nchunk * c * 2;
  *S = *AB + (size_t)n * c * 2;
  return bwd_reduce_impl(da, x, stats, gamma, beta, *S, nullptr, nullptr, workspace, workspace_bytes, n, hw, c, groups, act, 0, false, stream);
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

// ---- conv_out(swish(norm_out(x))) backward: see convout_bwd_kernel ----
static int convout_geom(Geom& g, int n, int h, int w, int c, int groups) {
  if (n <= 0 || h <= 0 || w <= 0 || c != 128 || groups <= 0 || c % groups != 0 || (c / groups) % 2 != 0 || w % 16 != 0) return -1;
  const long long hw = (long long)h * w;
  if (hw * c * n >= (1ll << 40) || hw >= (1ll << 30)) return -1;
  g.HW = (int)hw; g.C = c; g.G = groups; g.cpg = c / groups; g.tp_shift = 4; g.rows = 16;
  int nchunk = (2048 + n - 1) / n;                    // ~2048 blocks, like make_geom
  int ppc = (int)((hw + nchunk - 1) / nchunk);
  ppc = (ppc + 15) / 16 * 16;
  if (ppc < 64) ppc = 64;
  g.ppc = ppc; g.nchunk = (int)((hw + ppc - 1) / ppc);
  return 0;
}
extern "C" int dmvae_norm_conv_out_bwd_supported(int n, int h, int w, int c, int groups, int cout) {
  Geom g;
  return (cout == 3 && convout_geom(g, n, h, w, c, groups) == 0) ? 1 : 0;
}
extern "C" size_t dmvae_norm_conv_out_bwd_workspace(int n, int h, int w, int c, int groups) {
  Geom g;
  if (convout_geom(g, n, h, w, c, groups)) return 0;
  return ((size_t)n * g.nchunk * c * 2 + (size_t)n * c * 2 + (size_t)n * groups * 2) * sizeof(float) + (size_t)n * (h + 2) * (w + 2) * 4 * sizeof(bf16);
}
extern "C" int dmvae_norm_conv_out_bwd(const void* dy, const void* w, const void* x, const void* stats, const void* gamma, const void* beta, void* dx, void* dgamma,
                                       void* dbeta, void* workspace, size_t workspace_bytes, int n, int h, int wd, int c, int groups, int cout, int accumulate,
                                       hipStream_t stream) {
  Geom g;
  DMVAE_CHECK_ARG(dy && w && x && stats && gamma && beta && dx && workspace, "norm_conv_out_bwd: null pointer");
  DMVAE_CHECK_ARG(cout == 3 && convout_geom(g, n, h, wd, c, groups) == 0,
                  "norm_conv_out_bwd: unsupported shape n=%d h=%d w=%d c=%d groups=%d cout=%d (c = 128, w %% 16 == 0, cout = 3)", n, h, wd, c, groups, cout);
  DMVAE_CHECK_ARG(workspace_bytes >= dmvae_norm_conv_out_bwd_workspace(n, h, wd, c, groups), "norm_conv_out_bwd: workspace too small");
  DMVAE_CHECK_ARG((dgamma == nullptr) == (dbeta == nullptr), "norm_conv_out_bwd: dgamma and dbeta go together");
  float* part = (float*)workspace;
  float* AB = part + (size_t)n * g.nchunk * c * 2;
  float* S = AB + (size_t)n * c * 2;
  bf16* dyp = (bf16*)(S + (size_t)n * groups * 2);       // 8-B pixels: every section before it is a multiple of 8 bytes
  const dim3 grid(g.nchunk, n);
  hipLaunchKernelGGL(convout_pad_kernel, dim3((unsigned)(((size_t)n * (h + 2) * (wd + 2) + 255) / 256)), dim3(256), 0, stream, (const float*)dy, dyp, n, h, wd);
  DMVAE_CHECK_LAUNCH();
  hipLaunchKernelGGL(convout_bwd_kernel<false>, grid, dim3(256), 0, stream, (const bf16*)dyp, (const float*)w, (const bf16*)x, (const float*)stats, (const float*)gamma,
                     (const float*)beta, part, (const float*)nullptr, (bf16*)nullptr, g, h, wd, (const float*)nullptr, (float*)nullptr, (float*)nullptr, n, 0);
  DMVAE_CHECK_LAUNCH();
  const int waves = n * groups;
  hipLaunchKernelGGL(bwd_final_kernel, dim3((waves + 3) / 4), dim3(256), 0, stream, part, (const float*)gamma, AB, S, g, n);
  DMVAE_CHECK_LAUNCH();
  hipLaunchKernelGGL(convout_bwd_kernel<true>, grid, dim3(256), 0, stream, (const bf16*)dyp, (const float*)w, (const bf16*)x, (const float*)stats, (const float*)gamma,
                     (const float*)beta, (float*)nullptr, (const float*)S, (bf16*)dx, g, h, wd, dgamma ? (const float*)AB : (const float*)nullptr, (float*)dgamma,
                     (float*)dbeta, n, accumulate);
  DMVAE_CHECK_LAUNCH();
  return 0;
}
