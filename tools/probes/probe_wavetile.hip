// Power / issue experiment for the conv GEMM core (no convolution semantics): the same 256x256 block tile, K streamed through LDS by LDS-DMA from an
// L2-resident operand image with random data, computed
//   A) as conv_pp.hip does it: 8 waves (2 x 4), 128x64 per wave (8 accumulators), 32-channel K tiles, two waves per SIMD alternating LOAD / COMPUTE;
//   B) 4 waves (2 x 2), 128x128 per wave (16 accumulators = 256 AGPRs), 64-channel K tiles, one wave per SIMD, fragments double-buffered in registers:
//      one third less LDS read traffic per MFMA, half the barriers.
// Prints sustained TFLOP/s for each (the chip is power-limited with random operands: DESIGN_HISTORY.md 3.5).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr unsigned SENT = 0x80000000u;
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ int swz64(int row) { return (row >> 2) & 3; }
__device__ __forceinline__ int swz128(int row) { return (row >> 1) & 7; }

constexpr int KROW = 4608;  // operand row length in elements (512 channels x 9 taps)

// ---------------------------------------------------------------- A: conv_pp's loop -----------------------------------------------------
__global__ __launch_bounds__(512) void kern_a(const bf16* wsrc, const bf16* xsrc, float* out, int reps) {
  constexpr int TM = 256, TP = 256, WP = 4, BM = 4, BP = 2, NBUF = 4, PF = 3;
  constexpr int TILE_A = TM * 64, SLOT = (TM + TP) * 64, NPA = 2, NPB = 2, NP = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), grp = wave >> 2;
  const int wm = wave / WP, wp = wave % WP;
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)wsrc, 0, 256u * KROW * 2u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)(xsrc + (size_t)blockIdx.x * 256 * KROW), 0, 256u * KROW * 2u, 0x00020000);
  unsigned voffA[NPA], voffB[NPB];
  for (int p = 0; p < NPA; p++) { const int row = (wave * NPA + p) * 16 + (lane >> 2); voffA[p] = (unsigned)row * KROW * 2u + (((lane & 3) ^ swz64(row)) << 4); }
  for (int p = 0; p < NPB; p++) { const int row = (wave * NPB + p) * 16 + (lane >> 2); voffB[p] = (unsigned)row * KROW * 2u + (((lane & 3) ^ swz64(row)) << 4); }
  const int kg = lane >> 5;
  int aoff[2][BM], boff[2][BP];
  for (int i = 0; i < BM; i++) { const int row = wm * 128 + i * 32 + (lane & 31); aoff[0][i] = row * 64 + ((kg ^ swz64(row)) << 4); aoff[1][i] = aoff[0][i] ^ 32; }
  for (int j = 0; j < BP; j++) { const int row = wp * 64 + j * 32 + (lane & 31); boff[0][j] = TILE_A + row * 64 + ((kg ^ swz64(row)) << 4); boff[1][j] = boff[0][j] ^ 32; }
  const int nK = KROW / 32;
  int it = 0;
  auto issue = [&](int slot) {
    const unsigned so = (unsigned)(it % nK) * 64u;
#pragma unroll
    for (int p = 0; p < NPA; p++) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, LPTR(smem + slot + (wave * NPA + p) * 1024), 16, voffA[p], so, 0, 0);
#pragma unroll
    for (int p = 0; p < NPB; p++) __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, LPTR(smem + slot + TILE_A + (wave * NPB + p) * 1024), 16, voffB[p], so, 0, 0);
    it++;
  };
  f32x16 acc[BM][BP];
  for (int i = 0; i < BM; i++) for (int j = 0; j < BP; j++) for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
#pragma unroll
  for (int u = 0; u < PF; u++) issue(u * SLOT);
  wait_vmcnt<(PF - 1) * NP>();
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();
  bf16x8 af[2][BM], bfr[2][BP];
  int slot_rd = 0, slot_wr = PF * SLOT;
  const int total = reps * nK;
#pragma unroll 1
  for (int t = 0; t < total; t++) {
    const char* sb = smem + slot_rd;
#pragma unroll
    for (int kk = 0; kk < 2; kk++) {
#pragma unroll
      for (int j = 0; j < BP; j++) bfr[kk][j] = *reinterpret_cast<const bf16x8*>(sb + boff[kk][j]);
#pragma unroll
      for (int i = 0; i < BM; i++) af[kk][i] = *reinterpret_cast<const bf16x8*>(sb + aoff[kk][i]);
    }
    issue(slot_wr);
    slot_rd = slot_rd + SLOT == NBUF * SLOT ? 0 : slot_rd + SLOT;
    slot_wr = slot_wr + SLOT == NBUF * SLOT ? 0 : slot_wr + SLOT;
    wait_vmcnt<(PF - 1) * NP>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; kk++)
#pragma unroll
      for (int i = 0; i < BM; i++)
#pragma unroll
        for (int j = 0; j < BP; j++) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(af[kk][i]), "v"(bfr[kk][j]));
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();
  wait_vmcnt<0>();
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  float s = 0.f;
  for (int i = 0; i < BM; i++) for (int j = 0; j < BP; j++) for (int r = 0; r < 16; r++) s += acc[i][j][r];
  out[(size_t)blockIdx.x * 512 + tid] = s;
}

// ---------------------------------------------------------------- D: A with 16x16x32 MFMAs (power comparison only: operand mapping not meaningful) ----
__global__ __launch_bounds__(512) void kern_d(const bf16* wsrc, const bf16* xsrc, float* out, int reps) {
  constexpr int TM = 256, TP = 256, WP = 4, BM = 4, BP = 2, NBUF = 4, PF = 3;
  constexpr int TILE_A = TM * 64, SLOT = (TM + TP) * 64, NPA = 2, NPB = 2, NP = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), grp = wave >> 2;
  const int wm = wave / WP, wp = wave % WP;
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)wsrc, 0, 256u * KROW * 2u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)(xsrc + (size_t)blockIdx.x * 256 * KROW), 0, 256u * KROW * 2u, 0x00020000);
  unsigned voffA[NPA], voffB[NPB];
  for (int p = 0; p < NPA; p++) { const int row = (wave * NPA + p) * 16 + (lane >> 2); voffA[p] = (unsigned)row * KROW * 2u + (((lane & 3) ^ swz64(row)) << 4); }
  for (int p = 0; p < NPB; p++) { const int row = (wave * NPB + p) * 16 + (lane >> 2); voffB[p] = (unsigned)row * KROW * 2u + (((lane & 3) ^ swz64(row)) << 4); }
  const int kg = lane >> 5;
  int aoff[2][BM], boff[2][BP];
  for (int i = 0; i < BM; i++) { const int row = wm * 128 + i * 32 + (lane & 31); aoff[0][i] = row * 64 + ((kg ^ swz64(row)) << 4); aoff[1][i] = aoff[0][i] ^ 32; }
  for (int j = 0; j < BP; j++) { const int row = wp * 64 + j * 32 + (lane & 31); boff[0][j] = TILE_A + row * 64 + ((kg ^ swz64(row)) << 4); boff[1][j] = boff[0][j] ^ 32; }
  const int nK = KROW / 32;
  int it = 0;
  auto issue = [&](int slot) {
    const unsigned so = (unsigned)(it % nK) * 64u;
#pragma unroll
    for (int p = 0; p < NPA; p++) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, LPTR(smem + slot + (wave * NPA + p) * 1024), 16, voffA[p], so, 0, 0);
#pragma unroll
    for (int p = 0; p < NPB; p++) __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, LPTR(smem + slot + TILE_A + (wave * NPB + p) * 1024), 16, voffB[p], so, 0, 0);
    it++;
  };
  f32x16 acc[BM][BP];
  for (int i = 0; i < BM; i++) for (int j = 0; j < BP; j++) for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
#pragma unroll
  for (int u = 0; u < PF; u++) issue(u * SLOT);
  wait_vmcnt<(PF - 1) * NP>();
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();
  bf16x8 af[2][BM], bfr[2][BP];
  int slot_rd = 0, slot_wr = PF * SLOT;
  const int total = reps * nK;
#pragma unroll 1
  for (int t = 0; t < total; t++) {
    const char* sb = smem + slot_rd;
#pragma unroll
    for (int kk = 0; kk < 2; kk++) {
#pragma unroll
      for (int j = 0; j < BP; j++) bfr[kk][j] = *reinterpret_cast<const bf16x8*>(sb + boff[kk][j]);
#pragma unroll
      for (int i = 0; i < BM; i++) af[kk][i] = *reinterpret_cast<const bf16x8*>(sb + aoff[kk][i]);
    }
    issue(slot_wr);
    slot_rd = slot_rd + SLOT == NBUF * SLOT ? 0 : slot_rd + SLOT;
    slot_wr = slot_wr + SLOT == NBUF * SLOT ? 0 : slot_wr + SLOT;
    wait_vmcnt<(PF - 1) * NP>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; kk++)
#pragma unroll
      for (int i = 0; i < BM; i++)
#pragma unroll
        for (int j = 0; j < BP; j++) {   // the same flops per K tile as two 32x32x16 per block pair, issued as four 16x16x32 on 4-register accumulators
          f32x4* sub = reinterpret_cast<f32x4*>(&acc[i][j]);
          asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(sub[kk * 2 + 0]) : "v"(af[kk][i]), "v"(bfr[kk][j]));
          asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(sub[kk * 2 + 1]) : "v"(af[kk ^ 1][i]), "v"(bfr[kk][j]));
        }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();
  wait_vmcnt<0>();
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  float s = 0.f;
  for (int i = 0; i < BM; i++) for (int j = 0; j < BP; j++) for (int r = 0; r < 16; r++) s += acc[i][j][r];
  out[(size_t)blockIdx.x * 512 + tid] = s;
}

// ---------------------------------------------------------------- C: A's schedule with 64-channel K tiles ------------------------------------
// 8 waves, two per SIMD alternating LOAD / COMPUTE, but a K tile is 64 channels (128-B LDS rows): 32 MFMAs per COMPUTE interval, half the barriers and half
// the per-tile issue overhead per MFMA; two 64-KiB slots (prefetch distance one tile).
__global__ __launch_bounds__(512) void kern_c(const bf16* wsrc, const bf16* xsrc, float* out, int reps) {
  constexpr int TM = 256, TP = 256, WP = 4, BM = 4, BP = 2;
  constexpr int TILE_A = TM * 128, SLOT = (TM + TP) * 128, NPA = 4, NPB = 4, NP = 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), grp = wave >> 2;
  const int wm = wave / WP, wp = wave % WP;
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)wsrc, 0, 256u * KROW * 2u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)(xsrc + (size_t)blockIdx.x * 256 * KROW), 0, 256u * KROW * 2u, 0x00020000);
  unsigned voffA[NPA], voffB[NPB];
  for (int p = 0; p < NPA; p++) { const int row = (wave * NPA + p) * 8 + (lane >> 3); voffA[p] = (unsigned)row * KROW * 2u + (((lane & 7) ^ swz128(row)) << 4); }
  for (int p = 0; p < NPB; p++) { const int row = (wave * NPB + p) * 8 + (lane >> 3); voffB[p] = (unsigned)row * KROW * 2u + (((lane & 7) ^ swz128(row)) << 4); }
  const int kg = lane >> 5;
  int arow[BM], brow[BP];
  for (int i = 0; i < BM; i++) arow[i] = wm * 128 + i * 32 + (lane & 31);
  for (int j = 0; j < BP; j++) brow[j] = wp * 64 + j * 32 + (lane & 31);
  const int nK = KROW / 64;
  int it = 0;
  auto issue = [&](int slot) {
    const unsigned so = (unsigned)(it % nK) * 128u;
#pragma unroll
    for (int p = 0; p < NPA; p++) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, LPTR(smem + slot + (wave * NPA + p) * 1024), 16, voffA[p], so, 0, 0);
#pragma unroll
    for (int p = 0; p < NPB; p++) __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, LPTR(smem + slot + TILE_A + (wave * NPB + p) * 1024), 16, voffB[p], so, 0, 0);
    it++;
  };
  f32x16 acc[BM][BP];
  for (int i = 0; i < BM; i++) for (int j = 0; j < BP; j++) for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
  issue(0);
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();
  bf16x8 af[4][BM], bfr[4][BP];
  int slot_rd = 0;
  const int total = reps * nK;
#pragma unroll 1
  for (int t = 0; t < total; t++) {
    const char* sb = smem + slot_rd;
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
#pragma unroll
      for (int j = 0; j < BP; j++) bfr[kk][j] = *reinterpret_cast<const bf16x8*>(sb + TILE_A + brow[j] * 128 + ((((kk << 1) | kg) ^ swz128(brow[j])) << 4));
#pragma unroll
      for (int i = 0; i < BM; i++) af[kk][i] = *reinterpret_cast<const bf16x8*>(sb + arow[i] * 128 + ((((kk << 1) | kg) ^ swz128(arow[i])) << 4));
    }
    // the other slot was last read two intervals ago by this group and one interval ago by the partner group, whose reads completed before the barrier
    // that ended its LOAD interval
    issue(slot_rd ^ SLOT);
    slot_rd ^= SLOT;
    wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 4; kk++)
#pragma unroll
      for (int i = 0; i < BM; i++)
#pragma unroll
        for (int j = 0; j < BP; j++) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(af[kk][i]), "v"(bfr[kk][j]));
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();
  wait_vmcnt<0>();
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  float s_ = 0.f;
  for (int i = 0; i < BM; i++) for (int j = 0; j < BP; j++) for (int r = 0; r < 16; r++) s_ += acc[i][j][r];
  out[(size_t)blockIdx.x * 512 + tid] = s_;
}

// ---------------------------------------------------------------- B: 4 waves, 128x128 per wave ------------------------------------------
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void kern_b(const bf16* wsrc, const bf16* xsrc, float* out, int reps) {
  constexpr int TM = 256, TP = 256, BM = 4, BP = 4, NBUF = 2;
  constexpr int TILE_A = TM * 128, SLOT = (TM + TP) * 128;       // 64-channel K tiles: 128-B rows, 64 KiB per slot
  constexpr int NPA = 8, NPB = 8, NP = 16;                       // 1-KiB pieces (8 rows x 128 B) per wave per K tile and operand
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wp = wave & 1;
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)wsrc, 0, 256u * KROW * 2u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)(xsrc + (size_t)blockIdx.x * 256 * KROW), 0, 256u * KROW * 2u, 0x00020000);
  unsigned voffA[NPA], voffB[NPB];
  for (int p = 0; p < NPA; p++) { const int row = (wave * NPA + p) * 8 + (lane >> 3); voffA[p] = (unsigned)row * KROW * 2u + (((lane & 7) ^ swz128(row)) << 4); }
  for (int p = 0; p < NPB; p++) { const int row = (wave * NPB + p) * 8 + (lane >> 3); voffB[p] = (unsigned)row * KROW * 2u + (((lane & 7) ^ swz128(row)) << 4); }
  const int kg = lane >> 5;
  int arow[BM], brow[BP];
  for (int i = 0; i < BM; i++) arow[i] = wm * 128 + i * 32 + (lane & 31);
  for (int j = 0; j < BP; j++) brow[j] = wp * 128 + j * 32 + (lane & 31);
  const int nK = KROW / 64;
  int it = 0;
  auto issue = [&](int slot) {
    const unsigned so = (unsigned)(it % nK) * 128u;
#pragma unroll
    for (int p = 0; p < NPA; p++) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, LPTR(smem + slot + (wave * NPA + p) * 1024), 16, voffA[p], so, 0, 0);
#pragma unroll
    for (int p = 0; p < NPB; p++) __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, LPTR(smem + slot + TILE_A + (wave * NPB + p) * 1024), 16, voffB[p], so, 0, 0);
    it++;
  };
  auto rdA = [&](const char* sb, int kk, int i) { return *reinterpret_cast<const bf16x8*>(sb + arow[i] * 128 + ((((kk << 1) | kg) ^ swz128(arow[i])) << 4)); };
  auto rdB = [&](const char* sb, int kk, int j) { return *reinterpret_cast<const bf16x8*>(sb + TILE_A + brow[j] * 128 + ((((kk << 1) | kg) ^ swz128(brow[j])) << 4)); };
  f32x16 acc[BM][BP];
  for (int i = 0; i < BM; i++) for (int j = 0; j < BP; j++) for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
  issue(0);
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  bf16x8 af[2][BM], bfr[2][BP];
  int slot_rd = 0;
#pragma unroll
  for (int i = 0; i < BM; i++) af[0][i] = rdA(smem, 0, i);
#pragma unroll
  for (int j = 0; j < BP; j++) bfr[0][j] = rdB(smem, 0, j);
  const int total = reps * nK;
#pragma unroll 1
  for (int t = 0; t < total; t++) {
    const char* sb = smem + slot_rd;
    const int slot_nx = slot_rd ^ SLOT;
    issue(slot_nx);  // tile t+1 into the other slot: its last readers (tile t-1) passed the barrier at the end of the previous iteration
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
      const int cur = kk & 1, nxt = cur ^ 1;
      if (kk < 3) {
#pragma unroll
        for (int i = 0; i < BM; i++) af[nxt][i] = rdA(sb, kk + 1, i);
#pragma unroll
        for (int j = 0; j < BP; j++) bfr[nxt][j] = rdB(sb, kk + 1, j);
      } else {
        wait_vmcnt<0>();                       // own pieces of tile t+1 landed
        __builtin_amdgcn_s_barrier();           // everybody's did, and everybody is done reading tile t
#pragma unroll
        for (int i = 0; i < BM; i++) af[nxt][i] = rdA(smem + slot_nx, 0, i);
#pragma unroll
        for (int j = 0; j < BP; j++) bfr[nxt][j] = rdB(smem + slot_nx, 0, j);
      }
#pragma unroll
      for (int i = 0; i < BM; i++)
#pragma unroll
        for (int j = 0; j < BP; j++) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(af[cur][i]), "v"(bfr[cur][j]));
    }
    slot_rd = slot_nx;
  }
  wait_vmcnt<0>();
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  float s = 0.f;
  for (int i = 0; i < BM; i++) for (int j = 0; j < BP; j++) for (int r = 0; r < 16; r++) s += acc[i][j][r];
  out[(size_t)blockIdx.x * 512 + tid] = s;
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 40, blocks = 256, zeros = argc > 2 ? atoi(argv[2]) : 0;
  const size_t nw = 256ull * KROW, nx = (size_t)blocks * 256 * KROW;
  std::vector<unsigned short> h(nx);
  srand(1);
  for (size_t i = 0; i < nx; i++) { float f = zeros ? 0.f : ((rand() & 0xffff) / 32768.f - 1.f); unsigned u; memcpy(&u, &f, 4); h[i] = (unsigned short)(u >> 16); }
  bf16 *w, *x; float* out;
  CK(hipMalloc(&w, nw * 2)); CK(hipMalloc(&x, nx * 2)); CK(hipMalloc(&out, blocks * 512 * 4));
  CK(hipMemcpy(w, h.data(), nw * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(x, h.data(), nx * 2, hipMemcpyHostToDevice));
  CK(hipFuncSetAttribute((const void*)kern_a, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 512 * 64));
  CK(hipFuncSetAttribute((const void*)kern_b, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 512 * 128));
  CK(hipFuncSetAttribute((const void*)kern_d, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 512 * 64));
  CK(hipFuncSetAttribute((const void*)kern_c, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 512 * 128));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double flop = (double)blocks * reps * 256.0 * 256.0 * KROW * 2.0;
  for (int round = 0; round < 3; round++) {
    for (int v = 0; v < 4; v++) {
      const int n = 20;
      for (int k = 0; k < 3; k++) { if (v == 0) hipLaunchKernelGGL(kern_a, dim3(blocks), dim3(512), 4 * 512 * 64, 0, w, x, out, reps); else if (v == 1) hipLaunchKernelGGL(kern_b, dim3(blocks), dim3(256), 2 * 512 * 128, 0, w, x, out, reps); else if (v == 2) hipLaunchKernelGGL(kern_c, dim3(blocks), dim3(512), 2 * 512 * 128, 0, w, x, out, reps); else hipLaunchKernelGGL(kern_d, dim3(blocks), dim3(512), 4 * 512 * 64, 0, w, x, out, reps); }
      CK(hipEventRecord(e0));
      for (int k = 0; k < n; k++) { if (v == 0) hipLaunchKernelGGL(kern_a, dim3(blocks), dim3(512), 4 * 512 * 64, 0, w, x, out, reps); else if (v == 1) hipLaunchKernelGGL(kern_b, dim3(blocks), dim3(256), 2 * 512 * 128, 0, w, x, out, reps); else if (v == 2) hipLaunchKernelGGL(kern_c, dim3(blocks), dim3(512), 2 * 512 * 128, 0, w, x, out, reps); else hipLaunchKernelGGL(kern_d, dim3(blocks), dim3(512), 4 * 512 * 64, 0, w, x, out, reps); }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("%s %s: %.1f us / launch, %.1f TFLOP/s\n", v == 0 ? "A 8 waves 128x64 pingpong K32" : (v == 1 ? "B 4 waves 128x128 regpipe K64" : (v == 2 ? "C 8 waves 128x64 pingpong K64" : "D = A with 16x16x32 MFMAs")), zeros ? "zeros" : "random", ms / n * 1e3, flop / (ms / n * 1e-3) / 1e12);
    }
  }
  CK(hipGetLastError());
  return 0;
}
