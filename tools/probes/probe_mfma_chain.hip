// Does the hardware interlock an accumulate chain through two DIFFERENT MFMA opcodes issued back to back?
//   v_mfma_f32_16x16x32_bf16 D, A, B, 0 ; v_mfma_f32_16x16x16_bf16 D, A2, B2, D
// (what hipcc 7.2 emitted, with no wait states in between, for csrc/groupnorm.hip::convout_bwd_kernel's first build).  Inline assembly, so the compiler adds and
// removes nothing.  mode 0: back to back; mode 1: s_nop 15 between; mode 2: the same-opcode chain (16x16x32 twice) back to back; reference: each product on its
// own (C = 0, s_nop 15 behind it), added on the VALU.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/probe_mfma_chain tools/probes/probe_mfma_chain.hip && /tmp/probe_mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k(const i32x4* A, const i32x4* B, const i32x4* A2, const i32x4* B2, f32x4* out, int mode) {
  const int l = threadIdx.x;
  i32x4 a = A[l], b = B[l], a2 = A2[l], b2 = B2[l];
  i32x2 a2h = {a2[0], a2[1]}, b2h = {b2[0], b2[1]};
  f32x4 d, r1, r2;
  if (mode == 0)
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0\n\tv_mfma_f32_16x16x16_bf16 %0, %3, %4, %0\n\ts_nop 15\n\ts_nop 15" : "=&v"(d) : "v"(a), "v"(b), "v"(a2h), "v"(b2h));
  else if (mode == 1)
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0\n\ts_nop 15\n\tv_mfma_f32_16x16x16_bf16 %0, %3, %4, %0\n\ts_nop 15\n\ts_nop 15" : "=&v"(d) : "v"(a), "v"(b), "v"(a2h), "v"(b2h));
  else if (mode == 2)
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0\n\tv_mfma_f32_16x16x32_bf16 %0, %3, %4, %0\n\ts_nop 15\n\ts_nop 15" : "=&v"(d) : "v"(a), "v"(b), "v"(a2), "v"(b2));
  else {
    if (mode == 3) {
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0\n\ts_nop 15\n\ts_nop 15" : "=&v"(r1) : "v"(a), "v"(b));
      asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, 0\n\ts_nop 15\n\ts_nop 15" : "=&v"(r2) : "v"(a2h), "v"(b2h));
    } else {
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0\n\ts_nop 15\n\ts_nop 15" : "=&v"(r1) : "v"(a), "v"(b));
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0\n\ts_nop 15\n\ts_nop 15" : "=&v"(r2) : "v"(a2), "v"(b2));
    }
    d = r1 + r2;
  }
  out[l] = d;
}

static unsigned short bf(float x) { unsigned u; memcpy(&u, &x, 4); return (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }

int main() {
  std::vector<unsigned short> h(4 * 64 * 8);
  srand(1);
  for (auto& v : h) v = bf((rand() / (float)RAND_MAX - 0.5f) * 2.f);
  void *A, *out;
  hipMalloc(&A, h.size() * 2); hipMalloc(&out, 64 * 16);
  hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  const i32x4* p = (const i32x4*)A;
  float res[5][256];
  for (int mode = 0; mode < 5; mode++) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, p, p + 64, p + 128, p + 192, (f32x4*)out, mode);
    hipMemcpy(res[mode], out, 64 * 16, hipMemcpyDeviceToHost);
  }
  auto diff = [&](int a, int b) { float m = 0, s = 0; for (int i = 0; i < 256; i++) { m = fmaxf(m, fabsf(res[a][i] - res[b][i])); s = fmaxf(s, fabsf(res[b][i])); } return m / s; };
  printf("16x16x32 -> 16x16x16 back to back      vs separate products: max rel diff %.3e\n", diff(0, 3));
  printf("16x16x32 -> s_nop 15 -> 16x16x16       vs separate products: max rel diff %.3e\n", diff(1, 3));
  printf("16x16x32 -> 16x16x32 back to back      vs separate products: max rel diff %.3e\n", diff(2, 4));
  return 0;
}
