// HBM-bound loss reductions of the DMVAE train step: coalesced 16-B loads, wave-shuffle + LDS block
// reduce, fixed-order second stage (deterministic; no float atomics).
//
//   l1_mse      train_tokenizer.py:180-181   (F.l1_loss, F.mse_loss) fused fwd + d/d recon
//   lpips_diff  utils/lpips.py:86-94,156-162 (normalize_tensor, diff^2, 1x1 lin, spatial mean) fwd + d/d feats1
//   dmd_pre/post train_dmd.py:204-230, toy_example_2d/dmd.py:349-360 (DMD score-gradient loss)
//   kl_mmd      build-defined (no reference counterpart)
#include "common.h"
#include "dmvae_hip.h"
#include <float.h>
#include <cstdlib>

namespace dmvae_loss {

__device__ __forceinline__ float block_sum_256(float v, float* sh) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// ---- L1 + MSE ---------------------------------------------------------------------------------
// part[b][0..1] = block sums of |d|, d^2 ; grad = w1*sign(d)/n + w2*2*d/n (if grad != null)
__global__ __launch_bounds__(256) void l1_mse_partial_kernel(const float* __restrict__ r, const float* __restrict__ x,
                                                             float* __restrict__ grad, float* __restrict__ part, size_t n4,
                                                             float g1, float g2) {
  __shared__ float sh[4];
  float s1 = 0.f, s2 = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const f32x4 a = reinterpret_cast<const f32x4*>(r)[i];
    const f32x4 b = reinterpret_cast<const f32x4*>(x)[i];
    f32x4 g;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const float d = a[e] - b[e];
      s1 += fabsf(d); s2 += d * d;
      g[e] = g1 * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) + g2 * d;
    }
    if (grad) reinterpret_cast<f32x4*>(grad)[i] = g;
  }
  s1 = block_sum_256(s1, sh);
  s2 = block_sum_256(s2, sh);
  if (threadIdx.x == 0) { part[blockIdx.x * 2] = s1; part[blockIdx.x * 2 + 1] = s2; }
}
__global__ void l1_mse_final_kernel(const float* __restrict__ part, float* __restrict__ out, int nb, double inv_n) {
  // single wave
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < nb; i += 64) { a += part[i * 2]; b += part[i * 2 + 1]; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
  if (threadIdx.x == 0) { out[0] = (float)(a * inv_n); out[1] = (float)(b * inv_n); }
}

// ---- LPIPS feature diff --------------------------------------------------------------------------
// feats NHWC bf16 [N][HW][C]; lpp = C/8 lanes per pixel.  part[n][chunk] = sum over the chunk's pixels of
// sum_c w_c (f0/(|f0|+eps) - f1/(|f1|+eps))^2 ; df1 (bf16, optional) = gscale * d(that)/d f1.
__global__ __launch_bounds__(256) void lpips_diff_kernel(const bf16* __restrict__ f0, const bf16* __restrict__ f1,
                                                         const float* __restrict__ w, bf16* __restrict__ df1,
                                                         float* __restrict__ part, int HW, int C, int lpp_shift, int ppc,
                                                         float gscale, float eps) {
  __shared__ float sh[4];
  const int lpp = 1 << lpp_shift;
  const int lane_c = threadIdx.x & (lpp - 1), prow = threadIdx.x >> lpp_shift;
  const int rows = 256 >> lpp_shift;
  const int n = blockIdx.y;
  const int p0 = blockIdx.x * ppc, p1 = min(p0 + ppc, HW);
  const bool active = lane_c * 8 < C;
  float wv[8];
#pragma unroll
  for (int e = 0; e < 8; e++) wv[e] = active ? w[lane_c * 8 + e] : 0.f;
  float acc = 0.f;
  // all lanes of a pixel group must iterate together (shuffles): loop bound depends on prow only
  for (int p = p0 + prow; p < p1; p += rows) {
    const size_t off = ((size_t)n * HW + p) * C + lane_c * 8;
    float a[8], b[8];
    if (active) {
      const bf16x8 va = dmvae_ldnt8(f0 + off);
      const bf16x8 vb = dmvae_ldnt8(f1 + off);
#pragma unroll
      for (int e = 0; e < 8; e++) { a[e] = (float)va[e]; b[e] = (float)vb[e]; }
    } else {
#pragma unroll
      for (int e = 0; e < 8; e++) { a[e] = 0.f; b[e] = 0.f; }
    }
    float sa = 0.f, sb = 0.f;
#pragma unroll
    for (int e = 0; e < 8; e++) { sa += a[e] * a[e]; sb += b[e] * b[e]; }
    for (int o = lpp >> 1; o > 0; o >>= 1) { sa += __shfl_xor(sa, o, 64); sb += __shfl_xor(sb, o, 64); }
    const float ra = sqrtf(sa), rb = sqrtf(sb);
    const float ia = 1.f / (ra + eps), ib = 1.f / (rb + eps);
    float g[8], v = 0.f, gdot = 0.f;
#pragma unroll
    for (int e = 0; e < 8; e++) {
#pragma clang fp contract(off)  // a*ia - b*ib must not become fma(a, ia, -(b*ib)): LPIPS(x, x) has to be exactly 0
      const float pa = a[e] * ia, pb = b[e] * ib;
      const float d = pa - pb;
      v += wv[e] * d * d;
      g[e] = -2.f * wv[e] * d;      // dL/d n1_c
      gdot += g[e] * b[e];
    }
    acc += v;
    if (df1) {
      for (int o = lpp >> 1; o > 0; o >>= 1) gdot += __shfl_xor(gdot, o, 64);
      const float k2 = rb > 0.f ? gdot * ib * ib / rb : 0.f;
      if (active) {
        bf16x8 o8;
#pragma unroll
        for (int e = 0; e < 8; e++) o8[e] = (bf16)(gscale * (g[e] * ib - b[e] * k2));
        *reinterpret_cast<bf16x8*>(df1 + off) = o8;
      }
    }
  }
  acc = block_sum_256(acc, sh);
  if (threadIdx.x == 0) part[(size_t)n * gridDim.x + blockIdx.x] = acc;
}
// The same level value and gradient, plus the 2x2 max pool of BOTH branches that follows every tapped level but the last in the VGG16 trunk (utils/lpips.py:126-135,
// _CFG "M"): the pool re-read the 2B feature maps the diff had just streamed (537 MB at relu1_2 with B = 32).  A pixel group walks pooled pixels and visits the four
// source pixels of each (all eight loads in flight first); per source pixel the arithmetic is lpips_diff_kernel's, the maximum is maxpool2x2_kernel's (first
// maximum in scan order).  pool0 / pool1: [N][H2][W2][C] halves of the pooled 2N-image tensor.
__global__ __launch_bounds__(256) void lpips_diff_pool_kernel(const bf16* __restrict__ f0, const bf16* __restrict__ f1, const float* __restrict__ w,
                                                              bf16* __restrict__ df1, float* __restrict__ part, bf16* __restrict__ pool0,
                                                              bf16* __restrict__ pool1, int H2, int W2, int C, int lpp_shift, int ppc, float gscale, float eps) {
  __shared__ float sh[4];
  const int lpp = 1 << lpp_shift;
  const int lane_c = threadIdx.x & (lpp - 1), prow = threadIdx.x >> lpp_shift;
  const int rows = 256 >> lpp_shift;
  const int n = blockIdx.y;
  const int HWp = H2 * W2;
  const int p0 = blockIdx.x * ppc, p1 = min(p0 + ppc, HWp);
  const bool active = lane_c * 8 < C;
  float wv[8];
#pragma unroll
  for (int e = 0; e < 8; e++) wv[e] = active ? w[lane_c * 8 + e] : 0.f;
  float acc = 0.f;
  const bf16x8 z8 = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
  for (int pp = p0 + prow; pp < p1; pp += rows) {
    const int yo = pp / W2, xo = pp - yo * W2;
    const size_t src = (((size_t)n * 2 * H2 + 2 * yo) * 2 * W2 + 2 * xo) * C + lane_c * 8;      // source pixel (2 yo, 2 xo)
    bf16x8 va[4], vb[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const size_t off = src + ((size_t)(k >> 1) * 2 * W2 + (k & 1)) * C;
      va[k] = active ? dmvae_ldnt8(f0 + off) : z8;
      vb[k] = active ? dmvae_ldnt8(f1 + off) : z8;
    }
    bf16x8 ma = va[0], mb = vb[0];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      float a[8], b[8];
#pragma unroll
      for (int e = 0; e < 8; e++) { a[e] = (float)va[k][e]; b[e] = (float)vb[k][e]; }
      if (k > 0) {
#pragma unroll
        for (int e = 0; e < 8; e++) { ma[e] = a[e] > (float)ma[e] ? va[k][e] : ma[e]; mb[e] = b[e] > (float)mb[e] ? vb[k][e] : mb[e]; }
      }
      float sa = 0.f, sb = 0.f;
#pragma unroll
      for (int e = 0; e < 8; e++) { sa += a[e] * a[e]; sb += b[e] * b[e]; }
      for (int o = lpp >> 1; o > 0; o >>= 1) { sa += __shfl_xor(sa, o, 64); sb += __shfl_xor(sb, o, 64); }
      const float ra = sqrtf(sa), rb = sqrtf(sb);
      const float ia = 1.f / (ra + eps), ib = 1.f / (rb + eps);
      float g[8], v = 0.f, gdot = 0.f;
#pragma unroll
      for (int e = 0; e < 8; e++) {
#pragma clang fp contract(off)  // as in lpips_diff_kernel: LPIPS(x, x) has to be exactly 0
        const float pa = a[e] * ia, pb = b[e] * ib;
        const float d = pa - pb;
        v += wv[e] * d * d;
        g[e] = -2.f * wv[e] * d;
        gdot += g[e] * b[e];
      }
      acc += v;
      if (df1) {
        for (int o = lpp >> 1; o > 0; o >>= 1) gdot += __shfl_xor(gdot, o, 64);
        const float k2 = rb > 0.f ? gdot * ib * ib / rb : 0.f;
        if (active) {
          bf16x8 o8;
#pragma unroll
          for (int e = 0; e < 8; e++) o8[e] = (bf16)(gscale * (g[e] * ib - b[e] * k2));
          *reinterpret_cast<bf16x8*>(df1 + src + ((size_t)(k >> 1) * 2 * W2 + (k & 1)) * C) = o8;
        }
      }
    }
    if (active) {
      const size_t po = ((size_t)n * HWp + pp) * C + lane_c * 8;
      *reinterpret_cast<bf16x8*>(pool0 + po) = ma;
      *reinterpret_cast<bf16x8*>(pool1 + po) = mb;
    }
  }
  acc = block_sum_256(acc, sh);
  if (threadIdx.x == 0) part[(size_t)n * gridDim.x + blockIdx.x] = acc;
}
// out[0] += scale * sum(part)   (scale = 1/(HW*N)); zero_first: overwrite
__global__ void scalar_sum_kernel(const float* __restrict__ part, float* __restrict__ out, int n, double scale, int accumulate) {
  double a = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) a += part[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
  if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.f) + (float)(a * scale);
}

// ---- DMD score-gradient loss ------------------------------------------------------------------------
// pre: xt = t*x1 + (1-t)*x0   (per-sample t)
__global__ void dmd_pre_kernel(const float* __restrict__ x1, const float* __restrict__ x0, const float* __restrict__ t,
                               float* __restrict__ xt, int per, size_t total) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const float tt = t[i / per];
    xt[i] = tt * x1[i] + (1.f - tt) * x0[i];
  }
}
__device__ __forceinline__ float nan_to_num_(float v) {
  if (v != v) return 0.f;
  if (v > FLT_MAX) return FLT_MAX;
  if (v < -FLT_MAX) return -FLT_MAX;
  return v;
}
// one block per sample; out_s[b] = {sum grad^2, ||grad||}; dlat = grad * gscale
__global__ __launch_bounds__(256) void dmd_post_kernel(const float* __restrict__ x1, const float* __restrict__ xt,
                                                       const float* __restrict__ t, const float* __restrict__ vtc,
                                                       const float* __restrict__ vtu, const float* __restrict__ vsc,
                                                       const float* __restrict__ vsu, float* __restrict__ dlat,
                                                       float* __restrict__ out_s, int per, float cfg, int use_wf, float gscale) {
  __shared__ float sh[4];
  const int b = blockIdx.x;
  const size_t base = (size_t)b * per;
  const float om = 1.f - t[b];
  const bool use_cfg = cfg > 1.f;
  float wf = 1.f;
  if (use_wf) {
    float s = 0.f;
    for (int i = threadIdx.x; i < per; i += 256) {
      float vt = vtc[base + i];
      if (use_cfg) vt = vt + (cfg - 1.f) * (vt - vtu[base + i]);
      const float pred_t = xt[base + i] + vt * om;
      s += fabsf(x1[base + i] - pred_t);
    }
    s = block_sum_256(s, sh);
    wf = s / (float)per;
  }
  float s2 = 0.f;
  for (int i = threadIdx.x; i < per; i += 256) {
    float vt = vtc[base + i], vs = vsc[base + i];
    if (use_cfg) { vt = vt + (cfg - 1.f) * (vt - vtu[base + i]); vs = vs + (cfg - 1.f) * (vs - vsu[base + i]); }
    const float x = xt[base + i], l = x1[base + i];
    const float p_real = l - (x + vt * om), p_stu = l - (x + vs * om);
    float g = p_real - p_stu;
    if (use_wf) g = g / wf;
    g = nan_to_num_(g);
    s2 += g * g;
    dlat[base + i] = g * gscale;
  }
  s2 = block_sum_256(s2, sh);
  if (threadIdx.x == 0) { out_s[b * 2] = s2; out_s[b * 2 + 1] = sqrtf(s2); }
}
// out[0] = 0.5 * sum(s2)/numel ; out[1] = mean_b ||grad_b||
__global__ void dmd_final_kernel(const float* __restrict__ out_s, float* __restrict__ out, int B, double inv_numel) {
  double a = 0.0, c = 0.0;
  for (int i = threadIdx.x; i < B; i += 64) { a += out_s[i * 2]; c += out_s[i * 2 + 1]; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); c += __shfl_xor(c, o, 64); }
  if (threadIdx.x == 0) { out[0] = (float)(0.5 * a * inv_numel); out[1] = (float)(c / B); }
}

// ---- per-latent KL of batch moments + batched RBF-mixture MMD (build-defined) -------------------------
//   kl[c]  = 0.5*(mu_c^2 + var_c - 1 - ln var_c), moments over all G*n rows of z [G][n][32]          (HBM-bound pass)
//   mmd[g] = mean k(x,x) + mean k(y,y) - 2 mean k(x,y),  k(a,b) = mean_j exp(-|a-b|^2 / (2*mult_j*d)), mult = {.5,1,2,4,8}
//   dz     = w_kl * d mean_c(kl_c)/dz + w_mmd * d mean_g(mmd_g)/dz
// Roofline note (DESIGN_HISTORY.md 3.4): compulsory traffic is (G*n + G*m + G*n)*32*4 B, but the pair work is G*(n^2+nm+m^2) kernel
// evaluations of ~46-78 VALU lane-ops each: at n = m = 256 the kernel is VALU/exp-bound, not HBM-bound; only the moments
// pass (and MMD with small groups) can approach the HBM roofline.
//
// Launches: kl_moments (grid-stride, 16-B loads) -> kl_final -> mmd_pair (one workgroup per (group, pair-type, 128-row
// tile)) -> kl_mmd_final.  Everything is fixed-order (no float atomics): run-to-run bit-exact.
constexpr int MMD_D = 32;
constexpr int MMD_ROWS = 128;   // rows per workgroup: 2 per lane
constexpr int MMD_CCH = 256;    // columns staged in LDS per chunk
constexpr int MMD_NQ = 4;       // column slices = waves per workgroup (2 workgroups share a CU: one stages / reduces while the other computes)

// part[b][c][0..1] = sum, sum of squares over the block's rows; z viewed as [R][32]
__global__ __launch_bounds__(256) void kl_moments_kernel(const float* __restrict__ z, float* __restrict__ part, size_t R) {
  __shared__ float red[32][8][8];
  const int cq = threadIdx.x & 7, rl = threadIdx.x >> 3;  // 8 lanes x float4 per row, 32 rows per sweep
  float s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
  const size_t stride = (size_t)gridDim.x * 32;
  size_t r = (size_t)blockIdx.x * 32 + rl;
  // The gradient pass walks z from its end right after this one; the last ~192 MB are therefore read with plain loads (they stay in the
  // 256 MB Infinity Cache for it), everything before with streaming loads (faster, and it would be evicted anyway).
  const size_t keep_rows = ((size_t)192 << 20) / (MMD_D * sizeof(float));
  const size_t nt_rows = R > keep_rows ? R - keep_rows : 0;
  constexpr int U = 4;
  for (; r + (U - 1) * stride < R; r += U * stride) {  // U independent 16-B loads in flight per lane
    f32x4 v[U];
    if (r + (U - 1) * stride < nt_rows) {
#pragma unroll
      for (int u = 0; u < U; u++) v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(z + (r + u * stride) * MMD_D + cq * 4));
    } else {
#pragma unroll
      for (int u = 0; u < U; u++) v[u] = *reinterpret_cast<const f32x4*>(z + (r + u * stride) * MMD_D + cq * 4);
    }
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
      for (int e = 0; e < 4; e++) { s[e] += v[u][e]; ss[e] += v[u][e] * v[u][e]; }
  }
  for (; r < R; r += stride) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(z + r * MMD_D + cq * 4);
#pragma unroll
    for (int e = 0; e < 4; e++) { s[e] += v[e]; ss[e] += v[e] * v[e]; }
  }
#pragma unroll
  for (int e = 0; e < 4; e++) { red[rl][cq][e] = s[e]; red[rl][cq][4 + e] = ss[e]; }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int c = threadIdx.x & 31, w = threadIdx.x >> 5;  // channel, 0: sum / 1: sum of squares
    float a = 0.f;
    for (int r = 0; r < 32; r++) a += red[r][c >> 2][w * 4 + (c & 3)];
    part[((size_t)blockIdx.x * MMD_D + c) * 2 + w] = a;
  }
}

// kl[c] from the moment partials; kl[32] = mean_c; stats[c] = (mu, var).  1024 threads: 16 strided part-lanes per (channel,
// moment), four independent loads in flight each (a serial loop over 2048 partials costs > 100 us of pure load latency)
__global__ __launch_bounds__(1024) void kl_final_kernel(const float* __restrict__ mom, float* __restrict__ kl, float* __restrict__ stats,
                                                        int nparts, double rows) {
  __shared__ double sh[16][64];
  const int cm = threadIdx.x & 63, pl = threadIdx.x >> 6;  // cm = channel*2 + moment (the partials' inner layout)
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  int g = pl;
  // sixteen independent loads in flight per lane (the same rows into the same four accumulators in the same order as four at a time: the kernel is one
  // block walking 512 KB of partials, i.e. pure load latency -- 14 us with four in flight, a tenth of the whole 268 MB KL pass)
  for (; g + 240 < nparts; g += 256) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; u++) v[u] = mom[(size_t)(g + 16 * u) * 64 + cm];
#pragma unroll
    for (int u = 0; u < 16; u += 4) { a0 += v[u]; a1 += v[u + 1]; a2 += v[u + 2]; a3 += v[u + 3]; }
  }
  for (; g + 48 < nparts; g += 64) {
    const float v0 = mom[(size_t)g * 64 + cm], v1 = mom[(size_t)(g + 16) * 64 + cm], v2 = mom[(size_t)(g + 32) * 64 + cm],
                v3 = mom[(size_t)(g + 48) * 64 + cm];
    a0 += v0; a1 += v1; a2 += v2; a3 += v3;
  }
  for (; g < nparts; g += 16) a0 += mom[(size_t)g * 64 + cm];
  sh[pl][cm] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (threadIdx.x < 64) {
    double t = 0.0;
    for (int k = 0; k < 16; k++) t += sh[k][threadIdx.x];
    sh[0][threadIdx.x] = t;
  }
  __syncthreads();
  if (threadIdx.x >= 64) return;
  const int c = threadIdx.x;
  double v = 0.0;
  if (c < MMD_D) {
    const double s = sh[0][2 * c], ss = sh[0][2 * c + 1];
    const double mu = s / rows;
    double var = ss / rows - mu * mu;
    if (var < 1e-30) var = 1e-30;
    v = 0.5 * (mu * mu + var - 1.0 - log(var));
    kl[c] = (float)v;
    stats[c * 2] = (float)mu; stats[c * 2 + 1] = (float)var;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if (c == 0) kl[MMD_D] = (float)(v / MMD_D);
}

// One workgroup = (group g, pair type, 128-row tile).  type 0: rows x / cols x (value + gradient), 1: rows x / cols y (value +
// gradient), 2: rows y / cols y (value).  Lane l of every wave holds rows l and l+64 of the tile in registers; wave q sweeps
// column slice q of the LDS-staged column chunk with wave-uniform (broadcast) ds_read_b128.
//   d2 = |a|^2 + |b|^2 - 2 a.b      k = mean_j e_j,  e_8 = exp(-d2/(16 d)), e_4 = e_8^2, ... (each halving of the bandwidth squares)
//   w  = mean_j e_j / (mult_j d)    (so that dk/da = -w (a - b));  per row: S = sum k, SW = sum w, SB = sum w b
// Outputs: ksum[g][tile] = sum of k over the tile (fixed order), gpart[type][g][row][:] = a*SW - SB = sum_j w (a - b_j).
template <bool GRAD>
__global__ __launch_bounds__(256, 2) void mmd_pair_kernel(const float* __restrict__ z, const float* __restrict__ y, float* __restrict__ ksum,
                                                       float* __restrict__ gpart, float* __restrict__ klpart, int n, int m, int tiles_x, int tiles_y,
                                                       int csplit) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* cols = sm;                          // [MMD_CCH][32]
  float* cnorm = sm + MMD_CCH * MMD_D;       // [MMD_CCH]
  constexpr int SCR = 2 * MMD_ROWS * 34 > MMD_CCH * MMD_D + MMD_CCH ? 2 * MMD_ROWS * 34 : MMD_CCH * MMD_D + MMD_CCH;
  float* sc = sm + SCR;                      // [MMD_NQ] tile-sum scalars, above both uses of the region below
  float* red = sm;                           // [2][MMD_ROWS][34] gradient scratch, aliases cols/cnorm once the sweep is over
  // csplit > 1 (small problems only): the columns of a (group, type, row tile) are shared out over `csplit` workgroups so that a 32-image batch still
  // puts two waves on every SIMD; each writes its own partial sums, combined in fixed order by kl_mmd_finish_kernel
  const int g = blockIdx.y, tile = blockIdx.x / csplit, cs = blockIdx.x - tile * csplit;
  const int type = tile < tiles_x ? 0 : (tile < 2 * tiles_x ? 1 : 2);
  const int rt = type == 0 ? tile : (type == 1 ? tile - tiles_x : tile - 2 * tiles_x);
  const float* rsrc = type == 2 ? y + (size_t)g * m * MMD_D : z + (size_t)g * n * MMD_D;
  const float* csrc = type == 0 ? z + (size_t)g * n * MMD_D : y + (size_t)g * m * MMD_D;
  const int nrows = type == 2 ? m : n, ncols_all = type == 0 ? n : m;
  const int cper = csplit > 1 ? ((ncols_all + csplit - 1) / csplit + 3) & ~3 : ncols_all;      // columns of this workgroup: [cbeg, cend)
  const int cbeg = min(cs * cper, ncols_all), ncols = min(cbeg + cper, ncols_all);
  const int lane = threadIdx.x & 63, q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r0 = rt * MMD_ROWS + lane, r1 = r0 + 64;

  float a0[MMD_D], a1[MMD_D], na0 = 0.f, na1 = 0.f;
#pragma unroll
  for (int e4 = 0; e4 < MMD_D / 4; e4++) {
    const f32x4 v0 = r0 < nrows ? *reinterpret_cast<const f32x4*>(rsrc + (size_t)r0 * MMD_D + e4 * 4) : f32x4{0, 0, 0, 0};
    const f32x4 v1 = r1 < nrows ? *reinterpret_cast<const f32x4*>(rsrc + (size_t)r1 * MMD_D + e4 * 4) : f32x4{0, 0, 0, 0};
#pragma unroll
    for (int e = 0; e < 4; e++) {
      a0[e4 * 4 + e] = v0[e]; a1[e4 * 4 + e] = v1[e];
      na0 += v0[e] * v0[e]; na1 += v1[e] * v1[e];
    }
  }
  // Fused path (klpart != null): the x-x blocks already hold their 128 rows of z in registers, in every wave -- wave q reduces channels 8q .. 8q+7
  // of this tile to (sum, sum of squares) for the KL moments, so that no separate pass over z is launched (rows past the end were loaded as 0).
  if (klpart && type == 0 && cs == 0) {
    float ms = 0.f, mss = 0.f;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const int c = q * 8 + e;
      float su = 0.f, sq = 0.f;
#pragma unroll
      for (int k = 0; k < MMD_D; k++)      // compile-time register index: select channel c = 8q + e without dynamic indexing
        if (k == c) { su = a0[k] + a1[k]; sq = a0[k] * a0[k] + a1[k] * a1[k]; }
      su = wave_sum(su); sq = wave_sum(sq);
      if (lane == e) { ms = su; mss = sq; }
    }
    if (lane < 8) {
      float* o = klpart + ((size_t)(g * tiles_x + rt) * MMD_D + q * 8 + lane) * 2;
      o[0] = ms; o[1] = mss;
    }
  }
  float s0 = 0.f, s1 = 0.f, sw0 = 0.f, sw1 = 0.f;
  float sb0[GRAD ? MMD_D : 1], sb1[GRAD ? MMD_D : 1];
  if (GRAD) {
#pragma unroll
    for (int e = 0; e < MMD_D; e++) { sb0[GRAD ? e : 0] = 0.f; sb1[GRAD ? e : 0] = 0.f; }
  }
  const float inv16d = 1.f / (16.f * MMD_D), invd = 1.f / (float)MMD_D;

  for (int c0 = cbeg; c0 < ncols; c0 += MMD_CCH) {
    const int cc = min(MMD_CCH, ncols - c0);
    // stage the column chunk: 8 lanes x float4 per column row, squared norms by an 8-lane shuffle
    for (int i = threadIdx.x; i < MMD_CCH * 8; i += 256) {
      const int j = i >> 3, e4 = i & 7;
      f32x4 v = {0, 0, 0, 0};
      if (j < cc) v = *reinterpret_cast<const f32x4*>(csrc + (size_t)(c0 + j) * MMD_D + e4 * 4);
      *reinterpret_cast<f32x4*>(cols + j * MMD_D + e4 * 4) = v;
      float p = v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
      p += __shfl_xor(p, 1, 64); p += __shfl_xor(p, 2, 64); p += __shfl_xor(p, 4, 64);
      if (e4 == 0) cnorm[j] = p;
    }
    __syncthreads();
    const int cpw = (cc + MMD_NQ - 1) / MMD_NQ;      // a full chunk: 64 columns per wave
    const int j1 = min(cc, (q + 1) * cpw);
    // software-pipelined column fetch: column j+1 is in flight (LDS broadcast reads) while column j is evaluated
    f32x4 bq[MMD_D / 4];
    float nbq = 0.f;
    {
      const int jf = min(q * cpw, MMD_CCH - 1);
#pragma unroll
      for (int e4 = 0; e4 < MMD_D / 4; e4++) bq[e4] = *reinterpret_cast<const f32x4*>(cols + jf * MMD_D + e4 * 4);
      nbq = cnorm[jf];
    }
    for (int j = q * cpw; j < j1; j++) {
      float b[MMD_D];
#pragma unroll
      for (int e4 = 0; e4 < MMD_D / 4; e4++)
#pragma unroll
        for (int e = 0; e < 4; e++) b[e4 * 4 + e] = bq[e4][e];
      const float nb = nbq;
      {
        const int jn = min(j + 1, MMD_CCH - 1);  // wave-uniform address: LDS broadcast
#pragma unroll
        for (int e4 = 0; e4 < MMD_D / 4; e4++) bq[e4] = *reinterpret_cast<const f32x4*>(cols + jn * MMD_D + e4 * 4);
        nbq = cnorm[jn];
      }
      float d0a = 0.f, d0b = 0.f, d1a = 0.f, d1b = 0.f;  // two partial chains per row for ILP
#pragma unroll
      for (int e = 0; e < MMD_D; e += 2) {
        d0a = fmaf(a0[e], b[e], d0a); d0b = fmaf(a0[e + 1], b[e + 1], d0b);
        d1a = fmaf(a1[e], b[e], d1a); d1b = fmaf(a1[e + 1], b[e + 1], d1b);
      }
      const float q0 = fmaxf(na0 + nb - 2.f * (d0a + d0b), 0.f), q1 = fmaxf(na1 + nb - 2.f * (d1a + d1b), 0.f);
      const float e8 = __expf(-q0 * inv16d), f8 = __expf(-q1 * inv16d);
      const float e4 = e8 * e8, e2 = e4 * e4, e1 = e2 * e2, eh = e1 * e1;
      const float f4 = f8 * f8, f2 = f4 * f4, f1 = f2 * f2, fh = f1 * f1;
      s0 += 0.2f * (e8 + e4 + e2 + e1 + eh);
      s1 += 0.2f * (f8 + f4 + f2 + f1 + fh);
      if (GRAD) {
        const float w0 = 0.2f * invd * (0.125f * e8 + 0.25f * e4 + 0.5f * e2 + e1 + 2.f * eh);
        const float w1 = 0.2f * invd * (0.125f * f8 + 0.25f * f4 + 0.5f * f2 + f1 + 2.f * fh);
        sw0 += w0; sw1 += w1;
#pragma unroll
        for (int e = 0; e < MMD_D; e++) { sb0[GRAD ? e : 0] = fmaf(w0, b[e], sb0[GRAD ? e : 0]); sb1[GRAD ? e : 0] = fmaf(w1, b[e], sb1[GRAD ? e : 0]); }
      }
    }
    __syncthreads();
  }
  if (r0 >= nrows) s0 = 0.f;
  if (r1 >= nrows) s1 = 0.f;
  // ---- tile sum of k: wave shuffle, then the 8 waves in fixed order ---------------------------------------------------------------
  float st = wave_sum(s0 + s1);
  if (lane == 0) sc[q] = st;
  // ---- per-row gradient sums across the 4 column slices: two scratch slots, two fixed-order rounds ---------------------------------
  if (GRAD && type != 2) {
    float* slot = red + (size_t)(q & 1) * MMD_ROWS * 34;
    if (q < 2) {
#pragma unroll
      for (int e = 0; e < MMD_D; e++) { slot[lane * 34 + e] = sb0[GRAD ? e : 0]; slot[(lane + 64) * 34 + e] = sb1[GRAD ? e : 0]; }
      slot[lane * 34 + 32] = sw0; slot[(lane + 64) * 34 + 32] = sw1;
    }
    __syncthreads();
    if (q >= 2) {
#pragma unroll
      for (int e = 0; e < MMD_D; e++) { slot[lane * 34 + e] += sb0[GRAD ? e : 0]; slot[(lane + 64) * 34 + e] += sb1[GRAD ? e : 0]; }
      slot[lane * 34 + 32] += sw0; slot[(lane + 64) * 34 + 32] += sw1;
    }
    __syncthreads();
    const int r = threadIdx.x >> 1, ch = threadIdx.x & 1;  // row, channel half
    const int rg = rt * MMD_ROWS + r;
    if (rg < nrows) {
      const float* s0p = red + r * 34;
      const float* s1p = red + (size_t)MMD_ROWS * 34 + r * 34;
      const float SW = s0p[32] + s1p[32];
      const float* ar = rsrc + (size_t)rg * MMD_D + ch * 16;
      float* o = gpart + ((((size_t)cs * 2 + type) * gridDim.y + g) * n + rg) * MMD_D + ch * 16;
#pragma unroll
      for (int e = 0; e < 16; e++) o[e] = ar[e] * SW - (s0p[ch * 16 + e] + s1p[ch * 16 + e]);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int k = 0; k < MMD_NQ; k++) t += sc[k];
    ksum[(size_t)g * gridDim.x + blockIdx.x] = t;      // [g][tile][cs]
  }
}

// mmd[g] from the tile sums (grid.x = G, 64 threads); dz from the gradient partials (grid-stride over G*n*8 float4s)
__global__ void mmd_final_kernel(const float* __restrict__ ksum, float* __restrict__ mmd, int n, int m, int tiles_x, int tiles_y) {
  const int g = blockIdx.x, nt = 2 * tiles_x + tiles_y;
  if (threadIdx.x != 0) return;
  double kxx = 0, kxy = 0, kyy = 0;
  for (int t = 0; t < tiles_x; t++) { kxx += ksum[(size_t)g * nt + t]; kxy += ksum[(size_t)g * nt + tiles_x + t]; }
  for (int t = 0; t < tiles_y; t++) kyy += ksum[(size_t)g * nt + 2 * tiles_x + t];
  mmd[g] = (float)(kxx / ((double)n * n) + kyy / ((double)m * m) - 2.0 * kxy / ((double)n * m));
}
__global__ __launch_bounds__(256) void kl_mmd_grad_kernel(const float* __restrict__ z, const float* __restrict__ gpart,
                                                          const float* __restrict__ stats, float* __restrict__ dz, int G, int n, int m,
                                                          float w_kl, float w_mmd) {
  const size_t total4 = (size_t)G * n * (MMD_D / 4);
  const float rows = (float)G * (float)n;
  // d mmd_g / dx_i = -(2/n^2) * gp_xx[i] + (2/(n m)) * gp_xy[i]   with gp = sum_j w (a - b_j)
  const float cxx = -2.f / ((float)n * (float)n) * w_mmd / (float)G, cxy = 2.f / ((float)n * (float)m) * w_mmd / (float)G;
  const float* gxx = gpart;
  const float* gxy = gpart + (size_t)G * n * MMD_D;
  // blockDim * gridDim is a multiple of 8, so a thread always owns the same channel quad: hoist its moments
  // The tensor is walked from its END: the moments pass has just streamed z front to back, so the tail is what the 256 MB Infinity
  // Cache still holds; total4 is a multiple of 8, so thread t then owns channel quad 7 - (t & 7) throughout.
  const int c4 = (7 - (int)(threadIdx.x & 7)) * 4;
  float mu[4], slope[4];
#pragma unroll
  for (int e = 0; e < 4; e++) {
    mu[e] = stats[(c4 + e) * 2];
    slope[e] = 1.f - 1.f / stats[(c4 + e) * 2 + 1];
  }
  const float kscale = w_kl / rows / (float)MMD_D;
  for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < total4; j += (size_t)gridDim.x * blockDim.x) {
    const size_t i = total4 - 1 - j;
    const f32x4 x = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(z) + i);
    f32x4 a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
    if (gpart) { a = reinterpret_cast<const f32x4*>(gxx)[i]; b = reinterpret_cast<const f32x4*>(gxy)[i]; }
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; e++) o[e] = kscale * (mu[e] + slope[e] * (x[e] - mu[e])) + cxx * a[e] + cxy * b[e];
    __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(dz) + i);
  }
}

// Matrix-core version of mmd_pair_kernel for the launch-bound training shapes (the two-launch path): same work split (one workgroup per group, pair type,
// 128-row tile and column share; wave q owns rows 32 q .. 32 q + 31), same outputs, but the two contractions over the 32 latent channels run on
// v_mfma_f32_32x32x2_f32 (true f32: the kernel values feed a difference of sums that cancels two digits) instead of 64 scalar FMAs per pair:
//   G^T[j][i] = b_j . a_i     A operand = columns from LDS (lane -> column j = l & 31, channel parity l >> 5), B operand = the wave's rows, in registers;
//                             result layout: lane <-> row i, registers <-> 16 of the block's 32 columns (j = 8 (r >> 2) + (r & 3) + 4 (l >> 5))
//   SB^T[c][i] += sum_j b_j[c] w_ij   the weights w are born in exactly the B-operand layout of this product (lane <-> i, one register <-> the two columns
//                             a K = 2 step consumes), so they go from the VALU straight back into the matrix pipe; A operand = b_j[c] from LDS.
// Per 32 x 32 pair block: 16 + 16 MFMAs and ~22 VALU operations per pair (norm combine, one exp, four squarings, the two bandwidth sums) that overlap them.
// Rows / columns past the end carry an infinite squared norm: every exp() is 0 there and no mask is needed.
// LDS columns: [j][parity][16] floats, 16-B slots XOR-ed by (j >> 1) & 7 (conflict-free 16-B operand reads across 16 consecutive columns).
// Extra workgroups past the pair blocks (fused KL): one per (group, x tile) sums the tile's rows and squares per channel (klpart).
__device__ __forceinline__ int mmd_slot(int j, int slot) { return j * 32 + ((slot ^ ((j >> 1) & 7)) << 2); }  // float index of a 16-B slot of column j

template <bool GRAD>
__global__ __launch_bounds__(256) void mmd_pair_mfma_kernel(const float* __restrict__ z, const float* __restrict__ y, float* __restrict__ ksum,
                                                            float* __restrict__ gpart, float* __restrict__ klpart, int n, int m, int tiles_x, int tiles_y,
                                                            int csplit, int npair) {
#if __HIP_DEVICE_COMPILE__
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int g = blockIdx.y, lane = threadIdx.x & 63, q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if ((int)blockIdx.x >= npair) {  // ---- KL moment partials of one 128-row x tile (fixed order) ----
    const int rt = blockIdx.x - npair;
    if (!klpart || rt >= tiles_x) return;
    float (*red)[8][8] = reinterpret_cast<float (*)[8][8]>(sm);  // [32 row lanes][8 channel quads][4 sums + 4 sums of squares]
    const int cq = threadIdx.x & 7, rl = threadIdx.x >> 3;
    float s4[4] = {0, 0, 0, 0}, ss4[4] = {0, 0, 0, 0};
    for (int r = rt * MMD_ROWS + rl; r < min(n, (rt + 1) * MMD_ROWS); r += 32) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(z + ((size_t)g * n + r) * MMD_D + cq * 4);
#pragma unroll
      for (int e = 0; e < 4; e++) { s4[e] += v[e]; ss4[e] = fmaf(v[e], v[e], ss4[e]); }
    }
#pragma unroll
    for (int e = 0; e < 4; e++) { red[rl][cq][e] = s4[e]; red[rl][cq][4 + e] = ss4[e]; }
    __syncthreads();
    if (threadIdx.x < 64) {
      const int c = threadIdx.x & 31, w = threadIdx.x >> 5;
      float a = 0.f;
      for (int r = 0; r < 32; r++) a += red[r][c >> 2][w * 4 + (c & 3)];
      klpart[((size_t)(g * tiles_x + rt) * MMD_D + c) * 2 + w] = a;
    }
    return;
  }
  const int tile = blockIdx.x / csplit, cs = blockIdx.x - tile * csplit;
  const int type = tile < tiles_x ? 0 : (tile < 2 * tiles_x ? 1 : 2);
  const int rt = type == 0 ? tile : (type == 1 ? tile - tiles_x : tile - 2 * tiles_x);
  const float* rsrc = type == 2 ? y + (size_t)g * m * MMD_D : z + (size_t)g * n * MMD_D;
  const float* csrc = type == 0 ? z + (size_t)g * n * MMD_D : y + (size_t)g * m * MMD_D;
  const int nrows = type == 2 ? m : n, ncols_all = type == 0 ? n : m;
  const int cper = csplit > 1 ? ((ncols_all + csplit - 1) / csplit + 31) & ~31 : (ncols_all + 31) & ~31;  // whole 32-column blocks per share
  const int cbeg = min(cs * cper, ncols_all), cend = min(cbeg + cper, ncols_all);
  const int nblk = (cend - cbeg + 31) >> 5;
  float* cols = sm;                         // [nblk * 32][32] in the parity-split, slot-swizzled layout
  float* cnorm = sm + (size_t)nblk * 32 * MMD_D;   // [nblk * 32], +inf past the end
  float* sc = cnorm + nblk * 32;            // [4] per-wave sums of k
  // ---- operands in: every global load of the block is issued before the first one is consumed (one memory round trip, not one per staging sweep: at
  // these sizes the kernel's duration is its latency chain) -----------------------------------------------------------------------------------------
  const int total = nblk * 32 * 8;          // 8 lanes x 16 B per column
  constexpr int SWEEPS = 8;                 // sweeps of 256 threads held in registers: shares of up to 256 columns
  f32x4 vst[SWEEPS];
#pragma unroll
  for (int it = 0; it < SWEEPS; it++) {
    const int i = threadIdx.x + it * 256, j = i >> 3, e4 = i & 7;
    vst[it] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (i < total && cbeg + j < cend) vst[it] = *reinterpret_cast<const f32x4*>(csrc + (size_t)(cbeg + j) * MMD_D + e4 * 4);
  }
  // this wave's 32 rows as the B operand: lane (i = l & 31, h = l >> 5) holds a_i[2 kk + h]
  const int il = lane & 31, h = lane >> 5;
  const int rg = rt * MMD_ROWS + q * 32 + il;
  f32x4 vrow[8];
#pragma unroll
  for (int k4 = 0; k4 < 8; k4++) {
    vrow[k4] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (rg < nrows) vrow[k4] = *reinterpret_cast<const f32x4*>(rsrc + (size_t)rg * MMD_D + k4 * 4);
  }
  // columns -> LDS: squared norms by an 8-lane butterfly; channels 4 e4 .. 4 e4 + 3 go to the parity halves (even ones to index 2 e4, 2 e4 + 1 of half 0)
  auto put = [&](int i, const f32x4& v) {
    const int j = i >> 3, e4 = i & 7;
    float p = v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    p += __shfl_xor(p, 1, 64); p += __shfl_xor(p, 2, 64); p += __shfl_xor(p, 4, 64);
    const int lo = mmd_slot(j, (2 * e4) >> 2) + ((2 * e4) & 3), hi = mmd_slot(j, 4 + ((2 * e4) >> 2)) + ((2 * e4) & 3);
    *reinterpret_cast<f32x2*>(cols + lo) = f32x2{v[0], v[2]};
    *reinterpret_cast<f32x2*>(cols + hi) = f32x2{v[1], v[3]};
    if (e4 == 0) cnorm[j] = cbeg + j < cend ? p : INFINITY;
  };
#pragma unroll
  for (int it = 0; it < SWEEPS; it++)
    if (threadIdx.x + it * 256 < total) put(threadIdx.x + it * 256, vst[it]);
  for (int i = threadIdx.x + SWEEPS * 256; i < total; i += 256) {  // larger shares: the remaining sweeps one by one
    const int j = i >> 3, e4 = i & 7;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (cbeg + j < cend) v = *reinterpret_cast<const f32x4*>(csrc + (size_t)(cbeg + j) * MMD_D + e4 * 4);
    put(i, v);
  }
  float areg[16];
  float na = 0.f;
#pragma unroll
  for (int k4 = 0; k4 < 8; k4++) {
    const f32x4 v = vrow[k4];
    areg[2 * k4] = h ? v[1] : v[0];
    areg[2 * k4 + 1] = h ? v[3] : v[2];
    na += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  if (rg >= nrows) na = INFINITY;
  __syncthreads();
  float s = 0.f, sw = 0.f;
  f32x16 sbt;
#pragma unroll
  for (int r = 0; r < 16; r++) sbt[r] = 0.f;
  const float inv16d = 1.f / (16.f * MMD_D), invd = 1.f / (float)MMD_D;
  // G^T of the next column block is issued before this block's VALU work (the matrix pipe runs it underneath).  An explicitly interleaved version (gradient
  // product of block jb - 1 and G^T of block jb + 1 alternating with the VALU work of block jb via sched_group_barrier) measured no faster: at the
  // training shape the launch is latency-, not issue-bound (DESIGN_HISTORY.md 3.4).
  auto gram = [&](int jb) {
    const int j = jb * 32 + il;
    f32x16 c;
#pragma unroll
    for (int r = 0; r < 16; r++) c[r] = 0.f;
#pragma unroll
    for (int s4 = 0; s4 < 4; s4++) {  // four 16-B operand reads cover this lane's 16 channels of column j
      const f32x4 bv = *reinterpret_cast<const f32x4*>(cols + mmd_slot(j, h * 4 + s4));
#pragma unroll
      for (int e = 0; e < 4; e++) c = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[e], areg[s4 * 4 + e], c, 0, 0, 0);
    }
    return c;
  };
  f32x16 acc_next;
#pragma unroll
  for (int r = 0; r < 16; r++) acc_next[r] = 0.f;
  if (nblk > 0) acc_next = gram(0);
  for (int jb = 0; jb < nblk; jb++) {
    const f32x16 acc = acc_next;
    if (jb + 1 < nblk) acc_next = gram(jb + 1);
    f32x16 wv;
#pragma unroll
    for (int r4 = 0; r4 < 4; r4++) {
      const f32x4 nb = *reinterpret_cast<const f32x4*>(cnorm + jb * 32 + 8 * r4 + 4 * h);
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const float d2 = fmaxf(na + nb[r] - 2.f * acc[r4 * 4 + r], 0.f);
        const float e8 = __expf(-d2 * inv16d);
        const float e4 = e8 * e8, e2 = e4 * e4, e1 = e2 * e2, eh = e1 * e1;
        s += 0.2f * (e8 + e4 + e2 + e1 + eh);
        if (GRAD) {
          const float w = 0.2f * invd * (0.125f * e8 + 0.25f * e4 + 0.5f * e2 + e1 + 2.f * eh);
          wv[r4 * 4 + r] = w;
          sw += w;
        }
      }
    }
    if (GRAD && type != 2) {
#pragma unroll
      for (int r = 0; r < 16; r++) {  // K step r: columns j_r(0) (lanes < 32) and j_r(0) + 4 (lanes >= 32); A operand = b_j[c], c = l & 31
        const int jr = jb * 32 + 8 * (r >> 2) + (r & 3) + 4 * h;
        const float bc = cols[mmd_slot(jr, (il & 1) * 4 + (il >> 3)) + ((il >> 1) & 3)];
        sbt = __builtin_amdgcn_mfma_f32_32x32x2f32(bc, wv[r], sbt, 0, 0, 0);
      }
    }
  }
  // ---- tile sum of k: wave sum, then the four waves in fixed order ---------------------------------------------------------------------------------
  const float st = wave_sum(s);
  if (lane == 0) sc[q] = st;
  if (GRAD && type != 2) {  // gpart[row][c] = a_i[c] * SW - SB[c]; lane (i, h) holds channels 8 r4 + 4 h + 0..3
    const float SW = sw + __shfl_xor(sw, 32, 64);
    if (rg < nrows) {
      const float* ar = rsrc + (size_t)rg * MMD_D;
      float* o = gpart + ((((size_t)cs * 2 + type) * gridDim.y + g) * n + rg) * MMD_D;
#pragma unroll
      for (int r4 = 0; r4 < 4; r4++) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(ar + 8 * r4 + 4 * h);
        f32x4 ov;
#pragma unroll
        for (int r = 0; r < 4; r++) ov[r] = av[r] * SW - sbt[r4 * 4 + r];
        *reinterpret_cast<f32x4*>(o + 8 * r4 + 4 * h) = ov;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) ksum[(size_t)g * npair + blockIdx.x] = (sc[0] + sc[1]) + (sc[2] + sc[3]);
#endif
}

// Second (and last) launch of the fused path: every block reduces the per-tile moment partials to the 32 channel statistics itself (nparts <= 1024
// rows of 64 floats, L2-resident, fixed order: every block arrives at bit-identical statistics), block 0 also writes kl[33], the statistics and
// mmd[g]; then the gradient pass of kl_mmd_grad_kernel.
__global__ __launch_bounds__(256) void kl_mmd_finish_kernel(const float* __restrict__ z, const float* __restrict__ gpart, const float* __restrict__ klpart,
                                                            int nparts, const float* __restrict__ ksum, float* __restrict__ kl, float* __restrict__ stats_out,
                                                            float* __restrict__ mmd, float* __restrict__ dz, int G, int n, int m, int tiles_x, int tiles_y,
                                                            int csplit, float w_kl, float w_mmd) {
  __shared__ double sh[4][64];
  __shared__ float st[64];
  const int cm = threadIdx.x & 63, pl = threadIdx.x >> 6;
  double a0 = 0.0, a1 = 0.0;
  int p = pl;
  for (; p + 4 < nparts; p += 8) { a0 += klpart[(size_t)p * 64 + cm]; a1 += klpart[(size_t)(p + 4) * 64 + cm]; }
  for (; p < nparts; p += 4) a0 += klpart[(size_t)p * 64 + cm];
  sh[pl][cm] = a0 + a1;
  __syncthreads();
  const double rows = (double)G * n;
  double v = 0.0;
  if (threadIdx.x < MMD_D) {
    const int c = threadIdx.x;
    const double s = (sh[0][2 * c] + sh[1][2 * c]) + (sh[2][2 * c] + sh[3][2 * c]);
    const double ss = (sh[0][2 * c + 1] + sh[1][2 * c + 1]) + (sh[2][2 * c + 1] + sh[3][2 * c + 1]);
    const double mu = s / rows;
    double var = ss / rows - mu * mu;
    if (var < 1e-30) var = 1e-30;
    st[2 * c] = (float)mu; st[2 * c + 1] = (float)var;
    v = 0.5 * (mu * mu + var - 1.0 - log(var));
    if (blockIdx.x == 0) { kl[c] = (float)v; stats_out[2 * c] = (float)mu; stats_out[2 * c + 1] = (float)var; }
  }
  if (blockIdx.x == 0) {
    if (threadIdx.x < 64) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      if (threadIdx.x == 0) kl[MMD_D] = (float)(v / MMD_D);
    }
    const int nt = (2 * tiles_x + tiles_y) * csplit;
    for (int g = threadIdx.x; g < G; g += 256) {
      double kxx = 0, kxy = 0, kyy = 0;
      for (int t = 0; t < tiles_x * csplit; t++) { kxx += ksum[(size_t)g * nt + t]; kxy += ksum[(size_t)g * nt + tiles_x * csplit + t]; }
      for (int t = 0; t < tiles_y * csplit; t++) kyy += ksum[(size_t)g * nt + 2 * tiles_x * csplit + t];
      mmd[g] = (float)(kxx / ((double)n * n) + kyy / ((double)m * m) - 2.0 * kxy / ((double)n * m));
    }
  }
  if (!dz) return;
  __syncthreads();
  const size_t total4 = (size_t)G * n * (MMD_D / 4);
  const float cxx = -2.f / ((float)n * (float)n) * w_mmd / (float)G, cxy = 2.f / ((float)n * (float)m) * w_mmd / (float)G;
  const float* gxx = gpart;
  const float* gxy = gpart + (size_t)G * n * MMD_D;
  const int c4 = (int)(threadIdx.x & 7) * 4;      // blockDim * gridDim is a multiple of 8: a thread keeps its channel quad
  float mu[4], slope[4];
#pragma unroll
  for (int e = 0; e < 4; e++) { mu[e] = st[(c4 + e) * 2]; slope[e] = 1.f - 1.f / st[(c4 + e) * 2 + 1]; }
  const float kscale = w_kl / ((float)G * (float)n) / (float)MMD_D;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
    const f32x4 x = reinterpret_cast<const f32x4*>(z)[i];
    f32x4 a = reinterpret_cast<const f32x4*>(gxx)[i], b = reinterpret_cast<const f32x4*>(gxy)[i];
    for (int k = 1; k < csplit; k++) {       // the column splits' partial gradients, fixed order
      const f32x4 ak = reinterpret_cast<const f32x4*>(gxx + (size_t)k * 2 * G * n * MMD_D)[i], bk = reinterpret_cast<const f32x4*>(gxy + (size_t)k * 2 * G * n * MMD_D)[i];
#pragma unroll
      for (int e = 0; e < 4; e++) { a[e] += ak[e]; b[e] += bk[e]; }
    }
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; e++) o[e] = kscale * (mu[e] + slope[e] * (x[e] - mu[e])) + cxx * a[e] + cxy * b[e];
    reinterpret_cast<f32x4*>(dz)[i] = o;
  }
}

}  // namespace dmvae_loss
using namespace dmvae_loss;

extern "C" size_t dmvae_loss_workspace(void) { return (size_t)2 * 4096 * sizeof(float) + 65536 * sizeof(float); }

extern "C" int dmvae_l1_mse(const void* recon, const void* images, void* grad, void* out2, void* workspace, size_t workspace_bytes,
                            size_t n, float w1, float w2, hipStream_t stream) {
  DMVAE_CHECK_ARG(recon && images && out2 && workspace, "l1_mse: null pointer");
  DMVAE_CHECK_ARG(n > 0 && n % 4 == 0, "l1_mse: element count must be a positive multiple of 4");
  DMVAE_CHECK_ARG(workspace_bytes >= dmvae_loss_workspace(), "l1_mse: workspace too small");
  size_t nb = (n / 4 + 255) / 256; if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(l1_mse_partial_kernel, dim3((unsigned)nb), dim3(256), 0, stream, (const float*)recon, (const float*)images, (float*)grad,
                     (float*)workspace, n / 4, w1 / (float)n, 2.f * w2 / (float)n);
  DMVAE_CHECK_LAUNCH();
  hipLaunchKernelGGL(l1_mse_final_kernel, dim3(1), dim3(64), 0, stream, (const float*)workspace, (float*)out2, (int)nb, 1.0 / (double)n);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

extern "C" int dmvae_lpips_diff(const void* f0, const void* f1, const void* lin_w, void* df1, void* out, void* workspace,
                                size_t workspace_bytes, int n, int hw, int c, float gscale, int accumulate, hipStream_t stream) {
  DMVAE_CHECK_ARG(f0 && f1 && lin_w && out && workspace, "lpips_diff: null pointer");
  DMVAE_CHECK_ARG(n > 0 && hw > 0 && c > 0 && c % 8 == 0 && c <= 512, "lpips_diff: c must be a multiple of 8 and <= 512 (got %d)", c);
  DMVAE_CHECK_ARG(workspace_bytes >= dmvae_loss_workspace(), "lpips_diff: workspace too small");
  int lpp = 1, sh = 0;
  while (lpp < c / 8) { lpp <<= 1; sh++; }
  const int rows = 256 / lpp;
  int nchunk = (2048 + n - 1) / n;
  const int maxc = (hw + rows - 1) / rows;
  if (nchunk > maxc) nchunk = maxc;
  if (nchunk * n > 65536) nchunk = 65536 / n;
  if (nchunk < 1) nchunk = 1;
  const int ppc = (hw + nchunk - 1) / nchunk;
  nchunk = (hw + ppc - 1) / ppc;
  DMVAE_CHECK_ARG((size_t)nchunk * n <= 65536, "lpips_diff: batch too large");
  float* part = (float*)workspace + 2 * 4096;
  hipLaunchKernelGGL(lpips_diff_kernel, dim3(nchunk, n), dim3(256), 0, stream, (const bf16*)f0, (const bf16*)f1, (const float*)lin_w,
                     (bf16*)df1, part, hw, c, sh, ppc, gscale, 1e-10f);
  DMVAE_CHECK_LAUNCH();
  hipLaunchKernelGGL(scalar_sum_kernel, dim3(1), dim3(64), 0, stream, part, (float*)out, nchunk * n, 1.0 / ((double)hw * n), accumulate);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

// dmvae_lpips_diff + the 2x2 max pool of both branches in one pass (lpips_diff_pool_kernel): h, w = the features' (even) height and width
extern "C" int dmvae_lpips_diff_pool(const void* f0, const void* f1, const void* lin_w, void* df1, void* out, void* pool0, void* pool1, void* workspace,
                                     size_t workspace_bytes, int n, int h, int w, int c, float gscale, int accumulate, hipStream_t stream) {
  DMVAE_CHECK_ARG(f0 && f1 && lin_w && out && pool0 && pool1 && workspace, "lpips_diff_pool: null pointer");
  DMVAE_CHECK_ARG(n > 0 && h > 0 && w > 0 && h % 2 == 0 && w % 2 == 0 && c > 0 && c % 8 == 0 && c <= 512 && (long long)h * w < (1ll << 30),
                  "lpips_diff_pool: h, w even, c a multiple of 8 and <= 512 (got h=%d w=%d c=%d)", h, w, c);
  DMVAE_CHECK_ARG(workspace_bytes >= dmvae_loss_workspace(), "lpips_diff_pool: workspace too small");
  int lpp = 1, sh = 0;
  while (lpp < c / 8) { lpp <<= 1; sh++; }
  const int rows = 256 / lpp, hwp = (h / 2) * (w / 2);
  int nchunk = (2048 + n - 1) / n;
  const int maxc = (hwp + rows - 1) / rows;
  if (nchunk > maxc) nchunk = maxc;
  if (nchunk * n > 65536) nchunk = 65536 / n;
  if (nchunk < 1) nchunk = 1;
  const int ppc = (hwp + nchunk - 1) / nchunk;
  nchunk = (hwp + ppc - 1) / ppc;
  DMVAE_CHECK_ARG((size_t)nchunk * n <= 65536, "lpips_diff_pool: batch too large");
  float* part = (float*)workspace + 2 * 4096;
  hipLaunchKernelGGL(lpips_diff_pool_kernel, dim3(nchunk, n), dim3(256), 0, stream, (const bf16*)f0, (const bf16*)f1, (const float*)lin_w, (bf16*)df1, part,
                     (bf16*)pool0, (bf16*)pool1, h / 2, w / 2, c, sh, ppc, gscale, 1e-10f);
  DMVAE_CHECK_LAUNCH();
  hipLaunchKernelGGL(scalar_sum_kernel, dim3(1), dim3(64), 0, stream, part, (float*)out, nchunk * n, 1.0 / ((double)h * w * n), accumulate);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

extern "C" int dmvae_dmd_pre(const void* x1, const void* x0, const void* t, void* xt, int batch, int per_sample, hipStream_t stream) {
  DMVAE_CHECK_ARG(x1 && x0 && t && xt && batch > 0 && per_sample > 0, "dmd_pre: bad argument");
  const size_t total = (size_t)batch * per_sample;
  size_t nb = (total + 255) / 256; if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(dmd_pre_kernel, dim3((unsigned)nb), dim3(256), 0, stream, (const float*)x1, (const float*)x0, (const float*)t, (float*)xt, per_sample, total);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

extern "C" int dmvae_dmd_post(const void* x1, const void* xt, const void* t, const void* v_teacher, const void* v_teacher_u,
                              const void* v_student, const void* v_student_u, void* dlatents, void* out2, void* workspace,
                              size_t workspace_bytes, int batch, int per_sample, float cfg, int weight_factor, hipStream_t stream) {
  DMVAE_CHECK_ARG(x1 && xt && t && v_teacher && v_student && dlatents && out2 && workspace, "dmd_post: null pointer");
  DMVAE_CHECK_ARG(cfg <= 1.f || (v_teacher_u && v_student_u), "dmd_post: cfg > 1 needs the unconditional outputs");
  DMVAE_CHECK_ARG(batch > 0 && batch <= 4096 && per_sample > 0, "dmd_post: bad shape");
  DMVAE_CHECK_ARG(workspace_bytes >= dmvae_loss_workspace(), "dmd_post: workspace too small");
  const double numel = (double)batch * per_sample;
  hipLaunchKernelGGL(dmd_post_kernel, dim3(batch), dim3(256), 0, stream, (const float*)x1, (const float*)xt, (const float*)t,
                     (const float*)v_teacher, (const float*)v_teacher_u, (const float*)v_student, (const float*)v_student_u,
                     (float*)dlatents, (float*)workspace, per_sample, cfg, weight_factor, (float)(1.0 / numel));
  DMVAE_CHECK_LAUNCH();
  hipLaunchKernelGGL(dmd_final_kernel, dim3(1), dim3(64), 0, stream, (const float*)workspace, (float*)out2, batch, 1.0 / numel);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

static inline void kl_mmd_plan(int groups, int n, int m, int* tx, int* ty, int* nmom) {
  *tx = (n + MMD_ROWS - 1) / MMD_ROWS;
  *ty = (m + MMD_ROWS - 1) / MMD_ROWS;
  size_t R = (size_t)groups * n;
  size_t nb = (R + 255) / 256;  // >= 8 row sweeps per block
  constexpr int cap = 1024;
  *nmom = (int)(nb > (size_t)cap ? (size_t)cap : (nb < 1 ? 1 : nb));
}
// Column split of the two-launch path: enough workgroups for two per CU (two waves per SIMD) on a 256-CU part, at most 4 splits; 1 otherwise.
static int g_dbg_fused = -1, g_dbg_csplit = -1, g_dbg_mfma = -1;    // dmvae_debug_kl_mmd: tests / A/B runs force a path; -1 = the planner's choice
static inline int kl_mmd_csplit(int groups, int tx, int ty) {
  if (g_dbg_csplit > 0) return g_dbg_csplit > 4 ? 4 : g_dbg_csplit;
  const long long blocks = (long long)groups * (2 * tx + ty);
  if ((size_t)groups * tx > 1024 || blocks >= 512) return 1;
  int cs = (int)((512 + blocks - 1) / blocks);
  return cs > 4 ? 4 : cs;
}
// layout (floats): mom[nmom][32][2] | stats[64] | ksum[G][2tx+ty][cs] | gpart[cs][2][G][n][32]
extern "C" size_t dmvae_kl_mmd_workspace(int groups, int n, int m) {
  if (groups <= 0 || n <= 0 || m < 0) return 0;
  int tx, ty, nmom;
  kl_mmd_plan(groups, n, m, &tx, &ty, &nmom);
  const size_t nmom_ws = (size_t)nmom > (size_t)groups * tx ? (size_t)nmom : (size_t)groups * tx;     // the fused path keeps one partial row per x tile
  const size_t cs = kl_mmd_csplit(groups, tx, ty);
  return (nmom_ws * MMD_D * 2 + 64 + cs * groups * (2 * tx + ty) + cs * 2 * groups * n * MMD_D) * sizeof(float);
}

extern "C" int dmvae_kl_mmd(const void* z, const void* y, void* kl, void* mmd, void* dz, void* workspace, size_t workspace_bytes,
                            int groups, int n, int m, int d, float w_kl, float w_mmd, hipStream_t stream) {
  const bool kl_only = m == 0;  // m = 0 (y, mmd may be NULL): the KL moment pass and its gradient alone
  DMVAE_CHECK_ARG(z && kl && workspace && (kl_only || (y && mmd)), "kl_mmd: null pointer");
  DMVAE_CHECK_ARG(d == MMD_D, "kl_mmd: latent width must be %d (got %d)", MMD_D, d);
  DMVAE_CHECK_ARG(groups > 0 && groups <= 65535 && n > 0 && m >= 0, "kl_mmd: need 0 < groups <= 65535, n > 0, m >= 0");
  const size_t need = dmvae_kl_mmd_workspace(groups, n, m);
  DMVAE_CHECK_ARG(workspace_bytes >= need, "kl_mmd: workspace too small (need %zu bytes, see dmvae_kl_mmd_workspace)", need);
  int tx, ty, nmom;
  kl_mmd_plan(groups, n, m, &tx, &ty, &nmom);
  const size_t nmom_ws = (size_t)nmom > (size_t)groups * tx ? (size_t)nmom : (size_t)groups * tx;
  float* mom = (float*)workspace;
  float* stats = mom + nmom_ws * MMD_D * 2;
  float* ksum = stats + 64;
  const int csplit_ws = kl_mmd_csplit(groups, tx, ty);
  float* gpart = ksum + (size_t)csplit_ws * groups * (2 * tx + ty);
  // Small problems (the training step's own shape: 32 images x 256 tokens) are launch-bound: two launches -- pair kernel with the KL moment partials
  // folded in, then one finishing kernel -- instead of five.  Large ones keep the streaming moment pass (HBM-bound, its own roofline) and the
  // single-block final reduce.  dmvae_debug_kl_mmd(0, ..) forces the five-launch path (tests).
  const bool fused_ok = g_dbg_fused != 0;
  const bool fused = fused_ok && !kl_only && (size_t)groups * tx <= 1024;
  if (!fused) {
    hipLaunchKernelGGL(kl_moments_kernel, dim3(nmom), dim3(256), 0, stream, (const float*)z, mom, (size_t)groups * n);
    DMVAE_CHECK_LAUNCH();
    hipLaunchKernelGGL(kl_final_kernel, dim3(1), dim3(1024), 0, stream, mom, (float*)kl, stats, nmom, (double)groups * n);
    DMVAE_CHECK_LAUNCH();
  }
  // cols + norms + tile scalars; the gradient scratch (2 x 128 x 34 floats = 34.8 KB) aliases cols + norms, the scalars sit above both
  if (kl_only) {
    if (dz) {
      size_t nb = ((size_t)groups * n * 8 + 255) / 256; if (nb > 4096) nb = 4096;
      hipLaunchKernelGGL(kl_mmd_grad_kernel, dim3((unsigned)nb), dim3(256), 0, stream, (const float*)z, (const float*)nullptr, stats, (float*)dz, groups, n, 1, w_kl, 0.f);
      DMVAE_CHECK_LAUNCH();
    }
    return 0;
  }
  const size_t lds_v = (size_t)(2 * MMD_ROWS * 34 + 8) * sizeof(float) > (size_t)(MMD_CCH * MMD_D + MMD_CCH + 8) * sizeof(float)
                           ? (size_t)(2 * MMD_ROWS * 34 + 8) * sizeof(float) : (size_t)(MMD_CCH * MMD_D + MMD_CCH + 8) * sizeof(float);
  const size_t lds_g = lds_v;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mmd_pair_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_g);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mmd_pair_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_v);
    attr_done = true;
  }
  const int csplit = fused ? csplit_ws : 1;
  const dim3 grid((2 * tx + ty) * csplit, groups);
  float* klpart = fused ? mom : nullptr;
  const bool mfma_ok = g_dbg_mfma != 0;   // 0: the scalar-FMA pair kernel (tests)
  if (fused && mfma_ok && n <= 1024 && m <= 1024) {  // launch-bound shapes: the two contractions of every pair on the matrix cores
    const int nmax = n > m ? n : m;
    const int cper = csplit > 1 ? ((nmax + csplit - 1) / csplit + 31) & ~31 : (nmax + 31) & ~31;
    const size_t lds = ((size_t)cper * (MMD_D + 1) + 8) * sizeof(float);
    const size_t lds_eff = lds < 8192 ? 8192 : lds;     // the moment blocks' reduction scratch
    static bool attr2 = false;
    if (!attr2) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mmd_pair_mfma_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mmd_pair_mfma_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
      attr2 = true;
    }
    const int npair = (2 * tx + ty) * csplit;
    const dim3 grid2(npair + tx, groups);           // + one moment block per x tile
    if (dz)
      hipLaunchKernelGGL(mmd_pair_mfma_kernel<true>, grid2, dim3(256), lds_eff, stream, (const float*)z, (const float*)y, ksum, gpart, klpart, n, m, tx, ty, csplit, npair);
    else
      hipLaunchKernelGGL(mmd_pair_mfma_kernel<false>, grid2, dim3(256), lds_eff, stream, (const float*)z, (const float*)y, ksum, gpart, klpart, n, m, tx, ty, csplit, npair);
    DMVAE_CHECK_LAUNCH();
  } else if (dz)
    hipLaunchKernelGGL(mmd_pair_kernel<true>, grid, dim3(256), lds_g, stream, (const float*)z, (const float*)y, ksum, gpart, klpart, n, m, tx, ty, csplit);
  else
    hipLaunchKernelGGL(mmd_pair_kernel<false>, grid, dim3(256), lds_v, stream, (const float*)z, (const float*)y, ksum, gpart, klpart, n, m, tx, ty, csplit);
  DMVAE_CHECK_LAUNCH();
  if (fused) {
    size_t nb = dz ? ((size_t)groups * n * 8 + 255) / 256 : 1;        // one float4 per thread: at 32 x 256 rows that is 256 workgroups, one per CU
    if (nb > 1024) nb = 1024;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(kl_mmd_finish_kernel, dim3((unsigned)nb), dim3(256), 0, stream, (const float*)z, gpart, klpart, groups * tx, ksum, (float*)kl, stats,
                       (float*)mmd, (float*)dz, groups, n, m, tx, ty, csplit, w_kl, w_mmd);
    DMVAE_CHECK_LAUNCH();
    return 0;
  }
  hipLaunchKernelGGL(mmd_final_kernel, dim3(groups), dim3(64), 0, stream, ksum, (float*)mmd, n, m, tx, ty);
  DMVAE_CHECK_LAUNCH();
  if (dz) {
    size_t nb = ((size_t)groups * n * 8 + 255) / 256; if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(kl_mmd_grad_kernel, dim3((unsigned)nb), dim3(256), 0, stream, (const float*)z, gpart, stats, (float*)dz, groups, n, m, w_kl, w_mmd);
    DMVAE_CHECK_LAUNCH();
  }
  return 0;
}

// diagnostics only (tests/test_gpu_kernels.py): fused = 0 forces the five-launch path, csplit > 0 the column split of the two-launch path, mfma = 0 the
// scalar-FMA pair kernel; -1 each = the planner's choice.  Results agree to f32 summation order either way.
extern "C" void dmvae_debug_kl_mmd(int fused, int csplit, int mfma) { g_dbg_fused = fused; g_dbg_csplit = csplit; g_dbg_mfma = mfma; }
