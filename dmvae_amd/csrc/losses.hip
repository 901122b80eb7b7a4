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
      const bf16x8 va = *reinterpret_cast<const bf16x8*>(f0 + off);
      const bf16x8 vb = *reinterpret_cast<const bf16x8*>(f1 + off);
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
      const float d = a[e] * ia - b[e] * ib;
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

// ---- fused per-latent KL moments + batched RBF-mixture MMD (build-defined) ---------------------------
// One block per group g (image): X_g = z[g] [n][d] (d == 32), Y_g = y[g] [m][d].
//   mom[g][c][0..1] = sum_r z, sum_r z^2   (KL batch moments, finalised by kl_final_kernel)
//   mmd[g] = mean k(x,x) + mean k(y,y) - 2 mean k(x,y),  k = mean_j exp(-|a-b|^2/(2*mult_j*d)), mult = {.5,1,2,4,8}
// Each thread owns one x row in registers (n <= 256) and streams y (and x) rows from LDS.
constexpr int MMD_D = 32;
__device__ __forceinline__ float rbf_mix(float d2, float inv16d) {
  // mult 8 -> exp(-d2/(16 d)); each halving of the bandwidth squares the kernel value
  const float e8 = __expf(-d2 * inv16d);
  const float e4 = e8 * e8, e2 = e4 * e4, e1 = e2 * e2, eh = e1 * e1;
  return 0.2f * (e8 + e4 + e2 + e1 + eh);
}
__global__ __launch_bounds__(256) void kl_mmd_kernel(const float* __restrict__ z, const float* __restrict__ y,
                                                     float* __restrict__ mom, float* __restrict__ mmd, int n, int m) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* xs = sm;                 // [n][33]
  float* ys = sm + (size_t)n * 33;  // [m][33]
  __shared__ float sh[4];
  const int g = blockIdx.x;
  const float* zg = z + (size_t)g * n * MMD_D;
  const float* yg = y + (size_t)g * m * MMD_D;
  for (int i = threadIdx.x; i < n * MMD_D; i += 256) xs[(i >> 5) * 33 + (i & 31)] = zg[i];
  for (int i = threadIdx.x; i < m * MMD_D; i += 256) ys[(i >> 5) * 33 + (i & 31)] = yg[i];
  __syncthreads();
  // KL moments: thread c<32 handles channel c (column sums over n rows); 8 row-slices x 32 channels
  {
    const int c = threadIdx.x & 31, sl = threadIdx.x >> 5;
    float s = 0.f, ss = 0.f;
    for (int r = sl; r < n; r += 8) { const float v = xs[r * 33 + c]; s += v; ss += v * v; }
    __shared__ float ms[8][32][2];
    ms[sl][c][0] = s; ms[sl][c][1] = ss;
    __syncthreads();
    if (threadIdx.x < 32) {
      float a = 0.f, b = 0.f;
      for (int k = 0; k < 8; k++) { a += ms[k][c][0]; b += ms[k][c][1]; }
      mom[((size_t)g * MMD_D + c) * 2] = a; mom[((size_t)g * MMD_D + c) * 2 + 1] = b;
    }
  }
  const float inv16d = 1.f / (16.f * MMD_D);
  float kxx = 0.f, kxy = 0.f, kyy = 0.f;
  const int i = threadIdx.x;
  float xi[MMD_D], yi[MMD_D];
  if (i < n) {
#pragma unroll
    for (int e = 0; e < MMD_D; e++) xi[e] = xs[i * 33 + e];
    for (int j = 0; j < n; j++) {
      float d2 = 0.f;
#pragma unroll
      for (int e = 0; e < MMD_D; e++) { const float d = xi[e] - xs[j * 33 + e]; d2 += d * d; }
      kxx += rbf_mix(d2, inv16d);
    }
    for (int j = 0; j < m; j++) {
      float d2 = 0.f;
#pragma unroll
      for (int e = 0; e < MMD_D; e++) { const float d = xi[e] - ys[j * 33 + e]; d2 += d * d; }
      kxy += rbf_mix(d2, inv16d);
    }
  }
  if (i < m) {
#pragma unroll
    for (int e = 0; e < MMD_D; e++) yi[e] = ys[i * 33 + e];
    for (int j = 0; j < m; j++) {
      float d2 = 0.f;
#pragma unroll
      for (int e = 0; e < MMD_D; e++) { const float d = yi[e] - ys[j * 33 + e]; d2 += d * d; }
      kyy += rbf_mix(d2, inv16d);
    }
  }
  kxx = block_sum_256(kxx, sh);
  kxy = block_sum_256(kxy, sh);
  kyy = block_sum_256(kyy, sh);
  if (threadIdx.x == 0) mmd[g] = kxx / ((float)n * n) + kyy / ((float)m * m) - 2.f * kxy / ((float)n * m);
}
// kl[c] = 0.5*(mu^2 + var - 1 - ln var) from moments over all G*n rows; kl[C] = mean_c ; also stats[c] = (mu, var)
__global__ void kl_final_kernel(const float* __restrict__ mom, float* __restrict__ kl, float* __restrict__ stats, int G, double rows) {
  const int c = threadIdx.x;  // 32 threads... launched with 64
  double v = 0.0;
  if (c < MMD_D) {
    double s = 0.0, ss = 0.0;
    for (int g = 0; g < G; g++) { s += mom[((size_t)g * MMD_D + c) * 2]; ss += mom[((size_t)g * MMD_D + c) * 2 + 1]; }
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
// dz[g][i][:] = w_kl * d(mean_c kl_c)/dz + w_mmd * d(mean_g mmd_g)/dz
__global__ __launch_bounds__(256) void kl_mmd_bwd_kernel(const float* __restrict__ z, const float* __restrict__ y,
                                                         const float* __restrict__ stats, float* __restrict__ dz, int n, int m,
                                                         int G, float w_kl, float w_mmd) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* xs = sm;
  float* ys = sm + (size_t)n * 33;
  const int g = blockIdx.x;
  const float* zg = z + (size_t)g * n * MMD_D;
  const float* yg = y + (size_t)g * m * MMD_D;
  for (int i = threadIdx.x; i < n * MMD_D; i += 256) xs[(i >> 5) * 33 + (i & 31)] = zg[i];
  for (int i = threadIdx.x; i < m * MMD_D; i += 256) ys[(i >> 5) * 33 + (i & 31)] = yg[i];
  __syncthreads();
  const int i = threadIdx.x;
  if (i >= n) return;
  const float inv16d = 1.f / (16.f * MMD_D);
  float xi[MMD_D], gr[MMD_D];
#pragma unroll
  for (int e = 0; e < MMD_D; e++) { xi[e] = xs[i * 33 + e]; gr[e] = 0.f; }
  // d k(a,b)/da = -(a-b) * sum_j (1/5) k_j / (mult_j * d) ; with mult = 8,4,2,1,.5
  const float cxx = 2.f / ((float)n * n), cxy = -2.f / ((float)n * m);
  for (int pass = 0; pass < 2; pass++) {
    const float* os = pass == 0 ? xs : ys;
    const int cnt = pass == 0 ? n : m;
    const float coef = pass == 0 ? cxx : cxy;
    for (int j = 0; j < cnt; j++) {
      float d[MMD_D], d2 = 0.f;
#pragma unroll
      for (int e = 0; e < MMD_D; e++) { d[e] = xi[e] - os[j * 33 + e]; d2 += d[e] * d[e]; }
      const float e8 = __expf(-d2 * inv16d);
      const float e4 = e8 * e8, e2 = e4 * e4, e1 = e2 * e2, eh = e1 * e1;
      const float kp = -0.2f * (e8 * 0.125f + e4 * 0.25f + e2 * 0.5f + e1 + eh * 2.f) / (float)MMD_D * coef;
#pragma unroll
      for (int e = 0; e < MMD_D; e++) gr[e] += kp * d[e];
    }
  }
  const double rows = (double)G * n;
#pragma unroll
  for (int e = 0; e < MMD_D; e++) {
    const float mu = stats[e * 2], var = stats[e * 2 + 1];
    const float dkl = (mu + (1.f - 1.f / var) * (xi[e] - mu)) / (float)rows / (float)MMD_D;
    dz[((size_t)g * n + i) * MMD_D + e] = w_kl * dkl + w_mmd * gr[e] / (float)G;
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

extern "C" int dmvae_kl_mmd(const void* z, const void* y, void* kl, void* mmd, void* dz, void* workspace, size_t workspace_bytes,
                            int groups, int n, int m, int d, float w_kl, float w_mmd, hipStream_t stream) {
  DMVAE_CHECK_ARG(z && y && kl && mmd && workspace, "kl_mmd: null pointer");
  DMVAE_CHECK_ARG(d == MMD_D, "kl_mmd: latent width must be %d (got %d)", MMD_D, d);
  DMVAE_CHECK_ARG(groups > 0 && n > 0 && n <= 256 && m > 0 && m <= 256, "kl_mmd: need 0 < n,m <= 256 per group");
  const size_t need = ((size_t)groups * MMD_D * 2 + MMD_D * 2) * sizeof(float);
  DMVAE_CHECK_ARG(workspace_bytes >= need, "kl_mmd: workspace too small (need %zu bytes)", need);
  float* mom = (float*)workspace;
  float* stats = mom + (size_t)groups * MMD_D * 2;
  const size_t lds = (size_t)(n + m) * 33 * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kl_mmd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 512 * 33 * 4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kl_mmd_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 512 * 33 * 4);
    attr_done = true;
  }
  hipLaunchKernelGGL(kl_mmd_kernel, dim3(groups), dim3(256), lds, stream, (const float*)z, (const float*)y, mom, (float*)mmd, n, m);
  DMVAE_CHECK_LAUNCH();
  hipLaunchKernelGGL(kl_final_kernel, dim3(1), dim3(64), 0, stream, mom, (float*)kl, stats, groups, (double)groups * n);
  DMVAE_CHECK_LAUNCH();
  if (dz) {
    hipLaunchKernelGGL(kl_mmd_bwd_kernel, dim3(groups), dim3(256), lds, stream, (const float*)z, (const float*)y, stats, (float*)dz, n, m, groups, w_kl, w_mmd);
    DMVAE_CHECK_LAUNCH();
  }
  return 0;
}
