// Optimiser tail on flat f32 buffers: global grad-norm, clip, AdamW, EMA in two launches.
// Replaces torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW(betas=(.9,.95), eps=1e-8, wd=.005) +
// update_ema's per-tensor Python loop (train_tokenizer.py:140-150,382,415-419,437).  HBM-bound:
// reads p,g,m,v,ema and writes p,m,v,ema once (36 B/param) + one 4 B/param norm pass.
#include "common.h"
#include "dmvae_hip.h"

namespace dmvae_optim {

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, float* __restrict__ part, size_t n) {
  __shared__ float sh[4];
  float s = 0.f;
  const size_t n4 = n / 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const f32x4 v = reinterpret_cast<const f32x4*>(g)[i];
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = g[n4 * 4 + threadIdx.x]; s += v * v; }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
// norm_out[0] = sqrt(prev_sq*use_prev + sum(part)), norm_out[1] = clip coefficient min(1, max_norm/(norm+1e-6)),
// norm_out[2] = running sum of squares (for multi-buffer norms)
__global__ void norm_final_kernel(const float* __restrict__ part, float* __restrict__ norm_out, int nb, float max_norm, int use_prev) {
  double a = 0.0;
  for (int i = threadIdx.x; i < nb; i += 64) a += part[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
  if (threadIdx.x == 0) {
    if (use_prev) a += (double)norm_out[2];
    const float nrm = (float)sqrt(a);
    norm_out[0] = nrm;
    const float c = max_norm / (nrm + 1e-6f);
    norm_out[1] = c < 1.f ? c : 1.f;
    norm_out[2] = (float)a;
  }
}

// One element's update: torch.optim.AdamW's arithmetic (decoupled decay, bias corrections, IEEE sqrt and division) + the EMA; shared by the vector body and the tail.
__device__ __forceinline__ void adamw_one(float& pi, const float g, float& mi, float& vi, float* ema_i, const float coef, const float lr, const float b1, const float b2,
                                          const float eps, const float wd, const float step, const float bc2_sqrt, const float decay) {
  const float gi = g * coef;
  pi = pi * (1.f - lr * wd);
  mi = b1 * mi + (1.f - b1) * gi;
  vi = b2 * vi + (1.f - b2) * gi * gi;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  pi -= step * mi / denom;
  if (ema_i) *ema_i = *ema_i * decay + pi * (1.f - decay);
}

// 16 B per lane and stream (the scalar form moved 4 B per lane: 4.8 TB/s over the student's 20 GB per step); the same per-element arithmetic, so the same bits.
__global__ __launch_bounds__(256) void adamw_ema_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, float* __restrict__ ema,
                                                        const float* __restrict__ clip, size_t n, float lr, float b1, float b2,
                                                        float eps, float wd, float bc1, float bc2_sqrt, float decay, bf16* __restrict__ shadow) {
  const float coef = clip ? clip[1] : 1.f;
  const float step = lr / bc1;
  const size_t n4 = n / 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    f32x4 pv = reinterpret_cast<const f32x4*>(p)[i], mv = reinterpret_cast<const f32x4*>(m)[i], vv = reinterpret_cast<const f32x4*>(v)[i];
    const f32x4 gv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g) + i);
    f32x4 ev = {0.f, 0.f, 0.f, 0.f};
    if (ema) ev = reinterpret_cast<const f32x4*>(ema)[i];
#pragma unroll
    for (int e = 0; e < 4; e++) {
      float pe = pv[e], me = mv[e], ve = vv[e], ee = ev[e];
      adamw_one(pe, gv[e], me, ve, ema ? &ee : nullptr, coef, lr, b1, b2, eps, wd, step, bc2_sqrt, decay);
      pv[e] = pe; mv[e] = me; vv[e] = ve; ev[e] = ee;
    }
    reinterpret_cast<f32x4*>(p)[i] = pv;
    reinterpret_cast<f32x4*>(m)[i] = mv;
    reinterpret_cast<f32x4*>(v)[i] = vv;
    if (ema) reinterpret_cast<f32x4*>(ema)[i] = ev;
    if (shadow) {   // the bf16 copy autocast would make of the new weight (RNE), for the next forward's GEMM operands
      const bf16x4 sv = {(bf16)pv[0], (bf16)pv[1], (bf16)pv[2], (bf16)pv[3]};
      reinterpret_cast<bf16x4*>(shadow)[i] = sv;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const size_t i = n4 * 4 + threadIdx.x;
    float pe = p[i], me = m[i], ve = v[i], ee = ema ? ema[i] : 0.f;
    adamw_one(pe, g[i], me, ve, ema ? &ee : nullptr, coef, lr, b1, b2, eps, wd, step, bc2_sqrt, decay);
    p[i] = pe; m[i] = me; v[i] = ve;
    if (ema) ema[i] = ee;
    if (shadow) shadow[i] = (bf16)pe;
  }
}

}  // namespace dmvae_optim
using namespace dmvae_optim;

extern "C" int dmvae_grad_norm(const void* grads, void* norm_out3, void* workspace, size_t workspace_bytes, size_t n, float max_norm,
                               int accumulate_prev, hipStream_t stream) {
  DMVAE_CHECK_ARG(grads && norm_out3 && workspace, "grad_norm: null pointer");
  DMVAE_CHECK_ARG(workspace_bytes >= 2048 * sizeof(float), "grad_norm: workspace too small");
  size_t nb = (n / 4 + 255) / 256; if (nb > 2048) nb = 2048; if (nb < 1) nb = 1;
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3((unsigned)nb), dim3(256), 0, stream, (const float*)grads, (float*)workspace, n);
  DMVAE_CHECK_LAUNCH();
  hipLaunchKernelGGL(norm_final_kernel, dim3(1), dim3(64), 0, stream, (const float*)workspace, (float*)norm_out3, (int)nb, max_norm, accumulate_prev);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

static int adamw_launch(void* params, const void* grads, void* exp_avg, void* exp_avg_sq, void* ema, void* shadow, const void* norm_out3, size_t n,
                        float lr, float beta1, float beta2, float eps, float weight_decay, int step, float ema_decay, hipStream_t stream) {
  DMVAE_CHECK_ARG(params && grads && exp_avg && exp_avg_sq, "adamw_ema_step: null pointer");
  DMVAE_CHECK_ARG(step >= 1, "adamw_ema_step: step counts from 1");
  if (n == 0) return 0;
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2 = 1.f - powf(beta2, (float)step);
  size_t nb = (n / 4 + 255) / 256; if (nb > 8192) nb = 8192; if (nb < 1) nb = 1;
  DMVAE_CHECK_ARG(((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq | (uintptr_t)ema) % 16 == 0 && (uintptr_t)shadow % 8 == 0,
                  "adamw_ema_step: buffers must be 16-byte aligned (8 for the bf16 shadow)");
  hipLaunchKernelGGL(adamw_ema_kernel, dim3((unsigned)nb), dim3(256), 0, stream, (float*)params, (const float*)grads, (float*)exp_avg,
                     (float*)exp_avg_sq, (float*)ema, (const float*)norm_out3, n, lr, beta1, beta2, eps, weight_decay, bc1, sqrtf(bc2), ema_decay,
                     (bf16*)shadow);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

extern "C" int dmvae_adamw_ema_step(void* params, const void* grads, void* exp_avg, void* exp_avg_sq, void* ema, const void* norm_out3,
                                    size_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                                    float ema_decay, hipStream_t stream) {
  return adamw_launch(params, grads, exp_avg, exp_avg_sq, ema, nullptr, norm_out3, n, lr, beta1, beta2, eps, weight_decay, step, ema_decay, stream);
}

extern "C" int dmvae_adamw_ema_step_shadow(void* params, const void* grads, void* exp_avg, void* exp_avg_sq, void* ema, void* bf16_shadow,
                                           const void* norm_out3, size_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                                           int step, float ema_decay, hipStream_t stream) {
  return adamw_launch(params, grads, exp_avg, exp_avg_sq, ema, bf16_shadow, norm_out3, n, lr, beta1, beta2, eps, weight_decay, step, ema_decay, stream);
}
