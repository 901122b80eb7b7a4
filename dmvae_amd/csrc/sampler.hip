// Downstream-consumer kernels: the per-step state update of the SDE sampler the reference's sample_50k.py runs
// (diffusion/transport/integrators.py:27-35 Euler-Maruyama step with the velocity -> score conversion of path.py:74-89 and the drift of
// transport.py:254-257 folded in) and the image -> uint8 conversion of sample_50k.py:151.  Both are single HBM passes.
//
// Arithmetic follows the reference's f32 elementwise graph operation by operation (no contraction into FMAs, IEEE divide), so for the same
// model output the state after a step is bit-identical to the PyTorch-CPU result; every coefficient depends on t only and arrives as a scalar
// the host computed in f32 the way the reference does.
#include "common.h"
#include "dmvae_hip.h"

namespace dmvae_sampler {

template <typename TV>
__global__ __launch_bounds__(256) void sde_euler_kernel(const float* __restrict__ x, const TV* __restrict__ v, const float* __restrict__ w,
                                                        float* __restrict__ x_out, float* __restrict__ mean_out, size_t n4, float rar, float var,
                                                        float diff, float dt, float sq2d, float sqdt) {
#pragma clang fp contract(off)
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 xv = reinterpret_cast<const float4*>(x)[i];
    float vv[4];
    if constexpr (sizeof(TV) == 2) {
      const bf16x4 t = reinterpret_cast<const bf16x4*>(v)[i];
#pragma unroll
      for (int j = 0; j < 4; j++) vv[j] = (float)t[j];
    } else {
      const float4 t = reinterpret_cast<const float4*>(v)[i];
      vv[0] = t.x; vv[1] = t.y; vv[2] = t.z; vv[3] = t.w;
    }
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
    float wv[4] = {0.f, 0.f, 0.f, 0.f};
    if (w) { const float4 t = reinterpret_cast<const float4*>(w)[i]; wv[0] = t.x; wv[1] = t.y; wv[2] = t.z; wv[3] = t.w; }
    float mo[4], xo[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const float score = (rar * vv[j] - xs[j]) / var;       // get_score_from_velocity
      const float drift = vv[j] + diff * score;               // sde_drift
      const float mean = xs[j] + drift * dt;
      mo[j] = mean;
      xo[j] = mean + sq2d * (wv[j] * sqdt);                   // mean_x + sqrt(2 diffusion) * (w sqrt(dt))
    }
    if (mean_out) reinterpret_cast<float4*>(mean_out)[i] = make_float4(mo[0], mo[1], mo[2], mo[3]);
    if (x_out) reinterpret_cast<float4*>(x_out)[i] = w ? make_float4(xo[0], xo[1], xo[2], xo[3]) : make_float4(mo[0], mo[1], mo[2], mo[3]);
  }
}

// y [npix][c_stride] f32 (NHWC, first c channels used) -> out [npix][c] uint8 = trunc(clamp(127.5 * s + 128, 0, 255)); s optionally rounded
// to bf16 first (what `.float()` of an autocast decoder output holds).  NaN -> 0 is not defined by the reference (float -> uint8 of NaN); here 0.
__global__ __launch_bounds__(256) void image_to_u8_kernel(const float* __restrict__ y, uint8_t* __restrict__ out, size_t npix, int c, int c_stride,
                                                          int round_bf16) {
#pragma clang fp contract(off)
  for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += (size_t)gridDim.x * 256) {
    for (int j = 0; j < c; j++) {
      float s = y[p * c_stride + j];
      if (round_bf16) s = (float)(bf16)s;
      float q = 127.5f * s + 128.0f;
      q = q < 0.f ? 0.f : (q > 255.f ? 255.f : q);            // NaN compares false twice and converts to 0 below
      out[p * c + j] = (uint8_t)(int)(q == q ? q : 0.f);
    }
  }
}

}  // namespace dmvae_sampler

extern "C" int dmvae_sde_euler_step(const void* x, const void* v, int v_is_bf16, const void* w, void* x_out, void* mean_out, size_t n, float rar,
                                    float var, float diff, float dt, float sqrt_2diff, float sqrt_dt, hipStream_t stream) {
  using namespace dmvae_sampler;
  DMVAE_CHECK_ARG(x && v && (x_out || mean_out) && n > 0 && n % 4 == 0, "sde_euler_step: bad argument (n %% 4 == 0 required, got %zu)", n);
  const size_t n4 = n / 4;
  const int grid = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
  if (v_is_bf16)
    hipLaunchKernelGGL(sde_euler_kernel<bf16>, dim3(grid), dim3(256), 0, stream, (const float*)x, (const bf16*)v, (const float*)w, (float*)x_out,
                       (float*)mean_out, n4, rar, var, diff, dt, sqrt_2diff, sqrt_dt);
  else
    hipLaunchKernelGGL(sde_euler_kernel<float>, dim3(grid), dim3(256), 0, stream, (const float*)x, (const float*)v, (const float*)w, (float*)x_out,
                       (float*)mean_out, n4, rar, var, diff, dt, sqrt_2diff, sqrt_dt);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

extern "C" int dmvae_image_to_u8(const void* y, void* out, size_t npix, int c, int c_stride, int round_bf16, hipStream_t stream) {
  using namespace dmvae_sampler;
  DMVAE_CHECK_ARG(y && out && npix > 0 && c > 0 && c <= c_stride, "image_to_u8: bad argument");
  const int grid = (int)((npix + 255) / 256 < 4096 ? (npix + 255) / 256 : 4096);
  hipLaunchKernelGGL(image_to_u8_kernel, dim3(grid), dim3(256), 0, stream, (const float*)y, (uint8_t*)out, npix, c, c_stride, round_bf16);
  DMVAE_CHECK_LAUNCH();
  return 0;
}
