// Probe: how soon after a buffer_store_dwordx4 may a VALU instruction overwrite its data VGPRs?
// hipcc (ROCm 7.2) pads a >64-bit MUBUF store followed by a VALU write of its data with wait states only when the scalar offset is NOT a register; the conv
// epilogue (soffset in an SGPR, next row's v_pk_add_f32 into the upper half of the store data as the very next instruction) stored the next row's values in
// dwords 2-3 of some lanes.  Here the store and the overwrite sit in ONE asm block on fixed registers (v100-v103), K s_nop slots apart, every wave of every CU
// doing the same; any 0xBAD0xxxx word in memory = the store read its data after the overwrite.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define SETUP "v_mov_b32 v100, %0\n\tv_mov_b32 v101, %1\n\tv_mov_b32 v102, %2\n\tv_mov_b32 v103, %3\n\ts_nop 7\n\t"
#define STORE_S "buffer_store_dwordx4 v[100:103], %4, %5, %6 offen nt\n\t"
#define STORE_0 "buffer_store_dwordx4 v[100:103], %4, %5, 0 offen nt\n\t"
#define OVER_HI "v_mov_b32 v103, 0xBAD0BAD3\n\tv_mov_b32 v102, 0xBAD0BAD2\n\tv_mov_b32 v101, 0xBAD0BAD1\n\tv_mov_b32 v100, 0xBAD0BAD0"
#define OVER_LO "v_mov_b32 v100, 0xBAD0BAD0\n\tv_mov_b32 v101, 0xBAD0BAD1\n\tv_mov_b32 v102, 0xBAD0BAD2\n\tv_mov_b32 v103, 0xBAD0BAD3"
#define OVER_PK "v_pk_mov_b32 v[102:103], v[104:105], v[104:105]\n\tv_pk_mov_b32 v[100:101], v[104:105], v[104:105]"
#define ARGS :: "v"(0x600D0000u + r), "v"(0x600D0001u), "v"(0x600D0002u), "v"(0x600D0003u), "v"(vo), "s"(rY), "s"(so) : "v100", "v101", "v102", "v103", "v104", "v105", "memory"
#define PK_INIT "v_mov_b32 v104, 0xBAD0BAD2\n\tv_mov_b32 v105, 0xBAD0BAD3\n\t"
#define CASE(KK, GAP) if constexpr (K == KK) { \
    if constexpr (MODE == 0) { if constexpr (SOFF) asm volatile(SETUP STORE_S GAP OVER_LO ARGS); else asm volatile(SETUP STORE_0 GAP OVER_LO ARGS); } \
    else if constexpr (MODE == 1) { if constexpr (SOFF) asm volatile(SETUP STORE_S GAP OVER_HI ARGS); else asm volatile(SETUP STORE_0 GAP OVER_HI ARGS); } \
    else { if constexpr (SOFF) asm volatile(PK_INIT SETUP STORE_S GAP OVER_PK ARGS); else asm volatile(PK_INIT SETUP STORE_0 GAP OVER_PK ARGS); } }
template <int K, bool SOFF, int MODE>
__global__ __launch_bounds__(512) void k(unsigned* out, const char* src, int reps, int dma) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 1 << 24, 0x00020000);
  const unsigned voff = (unsigned)(((blockIdx.x * 8 + wave) * reps) * 1024 + (lane >> 3) * 128 + (lane & 7) * 16);  // 8 rows x 128 B per instruction
  for (int r = 0; r < reps; r++) {
    for (int p = 0; p < dma; p++)  // LDS-DMA pieces queued in front of the store (the next tile's prefetch)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rS, (__attribute__((address_space(3))) void*)(smem + (wave * 8 + (p & 7)) * 1024), 16,
                                               (unsigned)(lane * 16 + ((blockIdx.x * 64 + r * 8 + p) & 4095) * 1024), 0, 0, 0);
    const unsigned so = SOFF ? (unsigned)__builtin_amdgcn_readfirstlane(r * 1024) : 0u;
    const unsigned vo = SOFF ? voff : voff + r * 1024;
    CASE(0, "") CASE(1, "s_nop 0\n\t") CASE(2, "s_nop 1\n\t") CASE(3, "s_nop 2\n\t") CASE(4, "s_nop 3\n\t") CASE(8, "s_nop 7\n\t") CASE(16, "s_nop 15\n\t")
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
template <int K, bool SOFF, int MODE>
void run(unsigned* d, const char* src, size_t words, int dma) {
  const int reps = 64, grid = 256;
  hipMemset(d, 0, words * 4);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<K, SOFF, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  hipLaunchKernelGGL((k<K, SOFF, MODE>), dim3(grid), dim3(512), 64 * 1024, 0, d, src, reps, dma);
  hipDeviceSynchronize();
  std::vector<unsigned> h((size_t)grid * 8 * reps * 256);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  size_t bad = 0, good = 0, lanes[64] = {0}, dw[4] = {0};
  for (size_t i = 0; i < h.size(); i++) {
    if ((h[i] & 0xFFFF0000u) == 0xBAD00000u) { bad++; lanes[((i % 256) / 32) * 8 + ((i % 32) / 4)]++; dw[i & 3]++; }
    else if ((h[i] & 0xFFFF0000u) == 0x600D0000u) good++;
  }
  static const char* modes[] = {"dword 0 first", "dword 3 first", "v_pk_mov 2-3 first"};
  printf("%-18s gap %2d, soffset %s, %d DMA pieces in front: %9zu clobbered dwords (good %zu of %zu); by dword %zu %zu %zu %zu", modes[MODE], K, SOFF ? "sgpr" : "0   ", dma, bad, good,
         h.size(), dw[0], dw[1], dw[2], dw[3]);
  if (bad) { printf("; lanes:"); for (int l = 0; l < 64; l++) if (lanes[l]) printf(" %d", l); }
  printf("\n");
}
int main() {
  unsigned* d; char* src;
  const size_t words = (size_t)256 * 8 * 64 * 256;
  hipMalloc(&d, words * 4); hipMalloc(&src, 1 << 24); hipMemset(src, 0, 1 << 24);
  for (int dma : {0, 8}) {
    run<0, false, 0>(d, src, words, dma); run<0, true, 0>(d, src, words, dma);
    run<0, false, 1>(d, src, words, dma); run<0, true, 1>(d, src, words, dma);
    run<0, false, 2>(d, src, words, dma); run<0, true, 2>(d, src, words, dma);
    run<1, true, 1>(d, src, words, dma); run<1, true, 2>(d, src, words, dma);
    run<2, true, 1>(d, src, words, dma); run<2, true, 2>(d, src, words, dma);
    run<3, true, 2>(d, src, words, dma); run<4, true, 2>(d, src, words, dma); run<8, true, 2>(d, src, words, dma); run<16, true, 2>(d, src, words, dma);
  }
  return 0;
}
