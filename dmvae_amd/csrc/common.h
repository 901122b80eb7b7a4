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
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// XCD-aware block order (MI355X: 8 XCDs with private L2s; the dispatcher places flat block b on XCD b % 8).
// Maps the flat dispatch index to a logical work index such that each XCD walks ONE contiguous range of logical
// indices in dispatch order, so blocks that are neighbours in logical order share an L2.  Bijective for any total.
// Placement is a speed assumption only; results do not depend on it.
__device__ __forceinline__ unsigned xcd_remap(unsigned flat, unsigned total) {
  const unsigned q = total >> 3, r = total & 7u, x = flat & 7u;
  const unsigned start = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return start + (flat >> 3);
}
