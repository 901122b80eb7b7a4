// Shared device/host helpers for the DMVAE gfx950 kernels (MI355X / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))

// 64 B of zeros in device memory: source for padded / out-of-range rows of LDS-DMA tiles.
static __device__ uint4 dmvae_zero_page[4];  // one copy per translation unit (no -fgpu-rdc)

// error plumbing for the C ABI (thread-local last error string)
void dmvae_set_error(const char* fmt, ...);
#define DMVAE_CHECK_ARG(cond, ...) do { if (!(cond)) { dmvae_set_error(__VA_ARGS__); return -22; } } while (0)
#define DMVAE_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { dmvae_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e_)); return -5; } } while (0)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float bf2f(bf16 v) { return (float)v; }
// v_rcp_f32 (1 ulp), not an IEEE division: `1.0f / y` compiles to v_div_scale x2 + v_rcp + four FMAs + v_div_fmas + v_div_fixup -- ten instructions per element in
// kernels (GroupNorm backward: 23 of ~37 VALU instructions per element were the two divisions' sequences) whose results are rounded to bf16 anyway.
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// Exact-form GELU 0.5 x (1 + erf(x / sqrt 2)) (nn.GELU() of timm's Mlp, reached through models/vae.py:47-53) with erf from Abramowitz & Stegun 7.1.26
// (|error| <= 1.5e-7): E = (a1 t + ... + a5 t^5) exp(-z^2), t = 1 / (1 + p z), z = |x| / sqrt 2, and 1 + erf(x / sqrt 2) = E for x < 0, 2 - E for x >= 0 --
// the negative tail is computed without the 1 - erf cancellation.  ~15 VALU instructions, two of them transcendental, no branches; libm's erff is ~45 with two
// exec-masked branches, which as the epilogue of the fc1 GEMM (33.7 M values per call on ViT-L at batch 32) cost more than the HBM-bound stand-alone kernel
// it was fused to replace (29 vs 24 us).  Every result is rounded to bf16 (2^-9 relative) by its users, 1000 x coarser than the approximation.  Explicit
// fmaf: the stand-alone kernel (vit_bwd.hip) and the GEMM epilogue (gemm_pp.hip) give the same bits.
__device__ __forceinline__ float dmvae_gelu_f(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
  float p = __builtin_fmaf(1.061405429f, t, -1.453152027f);
  p = __builtin_fmaf(p, t, 1.421413741f);
  p = __builtin_fmaf(p, t, -0.284496736f);
  p = __builtin_fmaf(p, t, 0.254829592f);
  const float e = p * t * __expf(-z * z);
  return 0.5f * x * (x < 0.f ? e : 2.0f - e);
}

// XCD-aware block order (MI355X: 8 XCDs with private L2s; the dispatcher places flat block b on XCD b % 8).
// Maps the flat dispatch index to a logical work index such that each XCD walks ONE contiguous range of logical
// indices in dispatch order, so blocks that are neighbours in logical order share an L2.  Bijective for any total.
// Placement is a speed assumption only; results do not depend on it.
__device__ __forceinline__ unsigned xcd_remap(unsigned flat, unsigned total) {
  const unsigned q = total >> 3, r = total & 7u, x = flat & 7u;
  const unsigned start = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return start + (flat >> 3);
}

// 16-B load of a tensor this kernel reads ONCE: non-temporal, so that a streaming pass does not push the conv kernels' re-read operands out of L2 / the Infinity
// Cache (csrc/groupnorm.hip: -0.46 ms per step and 7-10 % on the passes themselves).
__device__ __forceinline__ bf16x8 dmvae_ldnt8(const bf16* p) {
  return __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p));
}
__device__ __forceinline__ unsigned dmvae_pack_bf16x2(float a, float b) {
  bf16x2 t = {(bf16)a, (bf16)b};
  return *reinterpret_cast<unsigned*>(&t);
}
// RMSNorm + RoPE of 8 consecutive channels d0..d0+7 of token `tok` (csrc/dit.hip::qknorm_rope_kernel's arithmetic: the normalised value is rounded to
// bf16 before the f32 weight multiplies; pairs (2i, 2i+1) rotate with their own table entries)
__device__ __forceinline__ uint4 dmvae_norm_rope8(const uint4 raw, float r, const float* __restrict__ w, const float* __restrict__ cosb,
                                            const float* __restrict__ sinb, int tok, int D, int d0) {
  const bf16x8 x = *reinterpret_cast<const bf16x8*>(&raw);
  const float4 w0 = *reinterpret_cast<const float4*>(w + d0), w1 = *reinterpret_cast<const float4*>(w + d0 + 4);
  const float4 c0 = *reinterpret_cast<const float4*>(cosb + (size_t)tok * D + d0), c1 = *reinterpret_cast<const float4*>(cosb + (size_t)tok * D + d0 + 4);
  const float4 s0 = *reinterpret_cast<const float4*>(sinb + (size_t)tok * D + d0), s1 = *reinterpret_cast<const float4*>(sinb + (size_t)tok * D + d0 + 4);
  const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
  const float cv[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
  const float sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
  float n[8];
#pragma unroll
  for (int e = 0; e < 8; e++) n[e] = (float)(bf16)((float)x[e] * r) * wv[e];
  uint4 o;
  o.x = dmvae_pack_bf16x2(n[0] * cv[0] - n[1] * sv[0], n[1] * cv[1] + n[0] * sv[1]);
  o.y = dmvae_pack_bf16x2(n[2] * cv[2] - n[3] * sv[2], n[3] * cv[3] + n[2] * sv[3]);
  o.z = dmvae_pack_bf16x2(n[4] * cv[4] - n[5] * sv[4], n[5] * cv[5] + n[4] * sv[5]);
  o.w = dmvae_pack_bf16x2(n[6] * cv[6] - n[7] * sv[6], n[7] * cv[7] + n[6] * sv[7]);
  return o;
}
__device__ __forceinline__ float dmvae_sumsq8(const uint4 raw) {
  const bf16x8 x = *reinterpret_cast<const bf16x8*>(&raw);
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 8; e++) s += (float)x[e] * (float)x[e];
  return s;
}

