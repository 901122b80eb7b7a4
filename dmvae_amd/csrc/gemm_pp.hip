// Linear-layer GEMM  Y[M][N] = act(X[M][K] . W[N][K]^T + bias)  for bf16 operands on gfx950 -- the transformer blocks' nn.Linear under autocast(bf16):
// timm's ViT blocks reached through models/vae.py:47-53 (qkv / proj / fc1 / fc2), diffusion/lightningdit/lightningdit.py:34-93,173-252 (qkv / proj),
// swiglu_ffn.py:15-36 (w12 / w3), and -- on a transposed bf16 copy of the weight -- their input gradients dX = dY . W.
//
// The loop is conv_pp.hip's ping-pong loop without the convolution: 8 waves, both operands K-contiguous and staged by LDS-DMA in K tiles of 32 (64-B LDS rows,
// XOR-swizzled 16-B chunks, conflict-free ds_read_b128 fragments for v_mfma_f32_16x16x32_bf16), a 4-deep ring filled three K tiles ahead under a counted vmcnt,
// the two waves of a SIMD alternating LOAD and COMPUTE intervals, persistent blocks (one per CU) walking an XCD-aware tile order, an LDS-staged epilogue that
// writes whole output rows.  What is new is the TILE: these GEMMs have few tiles per CU (M = batch x tokens = 8224 for ViT-L at batch 32 is 32.125 tiles of
// 256 rows, N = 1024 is four tiles of 256 columns: 132 tiles for 256 CUs), so what a call costs is decided by tile quantisation -- the vendor library's picks
// measure exactly "rounds x one 256 x 256 tile" (DESIGN.md 8.12).  Here the tile height is a template parameter in steps of 32 rows (wave grid 4 x 2, each wave
// 64 columns x 16 * BP16 rows) or 64 rows (wave grid 2 x 4), and the host picks, per (M, N, K), the instantiation whose round count x tile cost is lowest
// (dmvae_gemm_pp_plan): e.g. proj (N = 1024) runs 208 tiles of 256 x 160 in ONE round instead of 132 of 256 x 256 on half the chip.
//
// Epilogue: + bias (f32, or bf16 the way autocast hands it to the library), optional exact GELU (nn.GELU() of timm's Mlp) applied to the bf16-rounded
// pre-activation -- bit-identical to the Linear followed by csrc/vit_bwd.hip::gelu_fwd_kernel -- with the pre-activation optionally stored as a second result
// (the training route saves it for the backward pass), or SiLU; bf16 or f32 result.  Rows past M / columns past N never leave the CU: every store's per-lane
// offset is out of the descriptor's range for them (the range check does not see the scalar offset, so validity never rides on it).
#include "common.h"
#include "dmvae_hip.h"
#include <cstdlib>
#include <type_traits>

#ifndef DMVAE_GEMM_EXP   // timing experiments (tools/probes/build_variant.sh): 1 = no epilogue stores, 2 = no staging and no stores, 4 = nt stores
#define DMVAE_GEMM_EXP 0
#endif

namespace dmvae_gemm_pp {

struct Args {
  const bf16* x;     // [M][lda]
  const bf16* w;     // [N][ldw]
  const void* bias;  // [N] f32 / bf16, or null
  void* y;           // [M][ldy] bf16 / f32
  bf16* y2;          // [M][ldy] bf16 or null: the pre-activation (act != 0)
  int M, N, K, lda, ldw, ldy;
  int act, bias_bf16;
  int ntn, total;    // column tiles, tiles
};

constexpr unsigned SENT = 0x80000000u;  // voffset beyond any descriptor's num_records: loads return zeros, stores are dropped

__device__ __forceinline__ int swz64(int row) { return (0 - (row >> 2)) & 3; }   // conv_pp.hip: the 64-B-row chunk key that keeps ds_read_b128 fragments conflict-free

template <int N>
__device__ __forceinline__ void wait_vmcnt() {   // through the builtin: the compiler's wait-count pass has to SEE the wait (conv_pp.hip)
  __builtin_amdgcn_s_waitcnt((N & 15) | 0x0F70 | ((N >> 4) << 14));
  asm volatile("" ::: "memory");
}

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }   // vit_bwd.hip::gelu_f

// TM: output columns (weight rows) per tile, TP: output rows (tokens) per tile, WM x WP: wave grid over (columns, rows).
template <int TM, int TP, int WM, int WP, bool OUT_F32>
__global__ __launch_bounds__(512) void gemm_pp_kernel(Args a) {
#if __HIP_DEVICE_COMPILE__
  constexpr int NBUF = 4, PF = NBUF - 1;
  constexpr int BM16 = TM / WM / 16, BP16 = TP / WP / 16;   // 16 x 16 accumulator blocks per wave
  static_assert(WM * WP == 8 && BM16 * 16 * WM == TM && BP16 * 16 * WP == TP && BM16 * BP16 <= 32, "8 waves, at most 32 accumulators each");
  constexpr int TILE_A = TM * 64, TILE_B = TP * 64, SLOT = TILE_A + TILE_B;
  constexpr int NPA = (TM / 16 + 7) / 8, NPB = (TP / 16 + 7) / 8, NP = NPA + NPB;   // 1-KiB DMA pieces (16 rows x 64 B) per wave and K tile; pieces past the tile go to the dump KiB
  constexpr int CW = BM16 * 16;                                // output columns per wave
  constexpr int LPRr = CW / 8, LPR = LPRr <= 4 ? 4 : (LPRr <= 8 ? 8 : 16);   // lanes per staged row (8 columns each), rounded up to a power of two
  constexpr int RPI = 64 / LPR, NI = 16 / RPI;                 // rows per store instruction, store instructions per 16-row block
  constexpr int ROWB = CW * 4 + 16;                            // padded f32 row of the staging region
  constexpr int EPI_OFF = 2 * SLOT, EPI_BYTES = 8 * 16 * ROWB; // staging region: behind ring slots 0-1, which take the next tile's first K tiles meanwhile
  constexpr int RING = NBUF * SLOT;
  constexpr int DUMP_OFF = RING > EPI_OFF + EPI_BYTES ? RING : EPI_OFF + EPI_BYTES;   // 1 KiB behind everything else; exists only when a piece can miss the tile
  constexpr unsigned ES = OUT_F32 ? 4u : 2u;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const int wm = wave / WP, wp = wave % WP;
  const int nK = a.K >> 5;

  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (unsigned)a.N * (unsigned)a.ldw * 2u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (unsigned)a.M * (unsigned)a.lda * 2u, 0x00020000);

  int m0 = 0, n0 = 0;
  unsigned voffA[NPA], voffB[NPB];
  int it = 0;   // next K tile to issue (wave-uniform)
  auto setup = [&](unsigned work) {
    const unsigned wid = xcd_remap(work, a.total);
    m0 = (int)(wid / a.ntn) * TP;
    n0 = (int)(wid % a.ntn) * TM;
    it = 0;
#pragma unroll
    for (int p = 0; p < NPA; p++) {
      const int row = (wave * NPA + p) * 16 + (lane >> 2);
      const int n = n0 + row;
      const int c = (lane & 3) ^ swz64(row);   // logical 16-B chunk this lane fetches: the LDS image stays lane-linear, the swizzle is on the source address
      voffA[p] = (row < TM && n < a.N) ? (unsigned)n * (unsigned)a.ldw * 2u + c * 16u : SENT;
    }
#pragma unroll
    for (int p = 0; p < NPB; p++) {
      const int row = (wave * NPB + p) * 16 + (lane >> 2);
      const int m = m0 + row;
      const int c = (lane & 3) ^ swz64(row);
      voffB[p] = (row < TP && m < a.M) ? (unsigned)m * (unsigned)a.lda * 2u + c * 16u : SENT;
    }
  };
  // this wave's pieces of K tile `it` into the ring slot at byte offset `slot`; past the last K tile all-zero pieces (they move no memory) keep the vmcnt bookkeeping uniform
  auto issue = [&](int slot) {
    const bool live = it < nK;
    const unsigned so = (unsigned)it * 64u;
#pragma unroll
    for (int p = 0; p < NPA; p++) {
      const int g = wave * NPA + p;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, LPTR(smem + (g * 16 < TM ? slot + g * 1024 : DUMP_OFF)), 16, live ? voffA[p] : SENT, so, 0, 0);
    }
#pragma unroll
    for (int p = 0; p < NPB; p++) {
      const int g = wave * NPB + p;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, LPTR(smem + (g * 16 < TP ? slot + TILE_A + g * 1024 : DUMP_OFF)), 16, live ? voffB[p] : SENT, so, 0, 0);
    }
    it++;
  };

  // fragment read offsets inside a slot: one 16-B read per lane covers a 16-row x 32-deep fragment
  int aoff[BM16], boff[BP16];
#pragma unroll
  for (int i = 0; i < BM16; i++) {
    const int row = wm * CW + i * 16 + (lane & 15);
    aoff[i] = row * 64 + (((lane >> 4) ^ swz64(row)) << 4);
  }
#pragma unroll
  for (int j = 0; j < BP16; j++) {
    const int row = wp * (TP / WP) + j * 16 + (lane & 15);
    boff[j] = TILE_A + row * 64 + (((lane >> 4) ^ swz64(row)) << 4);
  }

  f32x4 acc[BM16][BP16];   // acc[i][j][r]: column block i, column 4 * (lane >> 4) + r; row block j, row lane & 15

  setup(blockIdx.x);
#pragma unroll
  for (int u = 0; u < PF; u++) issue(u * SLOT);

  for (unsigned work = blockIdx.x; work < (unsigned)a.total;) {
    const int m0c = m0, n0c = n0;
#pragma unroll
    for (int i = 0; i < BM16; i++)
#pragma unroll
      for (int j = 0; j < BP16; j++)
#pragma unroll
        for (int r = 0; r < 4; r++) acc[i][j][r] = 0.f;
    wait_vmcnt<(PF - 1) * NP>();
    __builtin_amdgcn_s_barrier();                  // B_0: everybody's pieces of K tile 0 have landed
    if (grp == 1) __builtin_amdgcn_s_setprio(1);   // static priority for the second-dispatched half (MI355X_MICROARCH.md, two waves per SIMD, item 4)
    if (grp == 1) __builtin_amdgcn_s_barrier();    // group 1 runs one interval behind group 0

    bf16x8 af[BM16], bfr[BP16];
    int slot_rd = 0, slot_wr = PF * SLOT;
#pragma unroll 1
    for (int t = 0; t < nK; t++) {
      // LOAD interval
      const char* sb = smem + slot_rd;
#pragma unroll
      for (int j = 0; j < BP16; j++) bfr[j] = *reinterpret_cast<const bf16x8*>(sb + boff[j]);
#pragma unroll
      for (int i = 0; i < BM16; i++) af[i] = *reinterpret_cast<const bf16x8*>(sb + aoff[i]);
      issue(slot_wr);
      slot_rd = slot_rd + SLOT == NBUF * SLOT ? 0 : slot_rd + SLOT;
      slot_wr = slot_wr + SLOT == NBUF * SLOT ? 0 : slot_wr + SLOT;
      wait_vmcnt<(PF - 1) * NP>();   // own pieces of the NEXT K tile have landed
      __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0) through the builtin: the compiler sees the fragments have arrived and puts no waits of its own between the MFMAs
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // COMPUTE interval
#pragma unroll
      for (int i = 0; i < BM16; i++)
#pragma unroll
        for (int j = 0; j < BP16; j++)
          asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(af[i]), "v"(bfr[j]));
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();    // matches group 1's extra barrier
    if (grp == 1) __builtin_amdgcn_s_setprio(0);
    wait_vmcnt<0>();                               // the trailing all-zero pieces
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // MFMA (inline asm, invisible to the hazard recogniser) -> accumulator reads
    __builtin_amdgcn_s_barrier();                  // every wave's trailing DMA has landed and all fragment reads are done: the ring is free

    // ---- epilogue: accumulators -> LDS (f32, per-wave region) -> whole output rows, 16-B stores ---------------------------------------------------------
    const unsigned next = work + gridDim.x;
    const bool has_next = next < (unsigned)a.total;
    {
      int lane_o = lane;
      asm volatile("" : "+v"(lane_o));   // recomputed per tile, not carried (spilled) across the main loop
      const int cl = lane_o % LPR, rg = lane_o / LPR;
      char* reg = smem + EPI_OFF + wave * (16 * ROWB);
      const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (unsigned)a.M * (unsigned)a.ldy * ES, 0x00020000);
      const __amdgpu_buffer_rsrc_t rY2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.y2, 0, a.y2 ? (unsigned)a.M * (unsigned)a.ldy * 2u : 0u, 0x00020000);
      const int cb = n0c + wm * CW + cl * 8;               // this lane's 8 output columns
      const bool c_ok = cl < LPRr && cb < a.N;
      // bias first, while the memory queue is empty (behind the next tile's prefetch its first use would wait for those pieces to land)
      f32x4 b_lo = {0.f, 0.f, 0.f, 0.f}, b_hi = {0.f, 0.f, 0.f, 0.f};
      {
        const __amdgpu_buffer_rsrc_t rBias =
            __builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0, a.bias ? (unsigned)a.N * (a.bias_bf16 ? 2u : 4u) : 0u, 0x00020000);
        if (a.bias_bf16) {
          const u32x4 b8 = __builtin_amdgcn_raw_buffer_load_b128(rBias, c_ok ? (unsigned)cb * 2u : SENT, 0, 0);
          const bf16x8 bb = *reinterpret_cast<const bf16x8*>(&b8);
          b_lo = f32x4{(float)bb[0], (float)bb[1], (float)bb[2], (float)bb[3]};
          b_hi = f32x4{(float)bb[4], (float)bb[5], (float)bb[6], (float)bb[7]};
        } else {
          const unsigned vo = c_ok ? (unsigned)cb * 4u : SENT;
          const u32x4 b0 = __builtin_amdgcn_raw_buffer_load_b128(rBias, vo, 0, 0), b1 = __builtin_amdgcn_raw_buffer_load_b128(rBias, vo + 16u, 0, 0);
          b_lo = *reinterpret_cast<const f32x4*>(&b0);
          b_hi = *reinterpret_cast<const f32x4*>(&b1);
        }
      }
      if (has_next) setup(next); else it = nK;
      issue(0);   // the next tile's first K tile goes out before the stores (unconditional: past the last tile all-zero pieces)
      const int mrow = m0c + wp * (TP / WP) + rg;          // the lane's row in store instruction 0 of row block 0
      auto body = [&](auto ACTc) __attribute__((always_inline)) {
        constexpr int ACT = decltype(ACTc)::value;
#pragma unroll
        for (int j = 0; j < BP16; j++) {
#if DMVAE_GEMM_EXP & 2
#pragma unroll
          for (int i = 0; i < BM16; i++) asm volatile("" :: "a"(acc[i][j]));
          continue;
#endif
#pragma unroll
          for (int i = 0; i < BM16; i++)
            *reinterpret_cast<f32x4*>(reg + (lane_o & 15) * ROWB + (i * 16 + 4 * (lane_o >> 4)) * 4) = acc[i][j];
          f32x4 lo[NI], hi[NI];
#pragma unroll
          for (int q = 0; q < NI; q++) {
            lo[q] = *reinterpret_cast<const f32x4*>(reg + (q * RPI + rg) * ROWB + cl * 32);
            hi[q] = *reinterpret_cast<const f32x4*>(reg + (q * RPI + rg) * ROWB + cl * 32 + 16);
          }
          u32x4 keep[OUT_F32 ? 2 * NI : NI], keep2[NI];
#pragma unroll
          for (int q = 0; q < NI; q++) {
            const f32x4 v0 = lo[q] + b_lo, v1 = hi[q] + b_hi;
            float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            const int m = mrow + j * 16 + q * RPI;
            const bool ok = c_ok && m < a.M;
            const unsigned eo = (unsigned)m * (unsigned)a.ldy + (unsigned)cb;
            if constexpr (ACT != 0) {
              // the activation sees the bf16-rounded pre-activation, as the unfused Linear -> activation pair does
              const u32x4 h = {dmvae_pack_bf16x2(v[0], v[1]), dmvae_pack_bf16x2(v[2], v[3]), dmvae_pack_bf16x2(v[4], v[5]), dmvae_pack_bf16x2(v[6], v[7])};
              keep2[q] = h;
              __builtin_amdgcn_raw_buffer_store_b128(h, rY2, ok ? eo * 2u : SENT, 0, 2);
              const bf16x8 hb = *reinterpret_cast<const bf16x8*>(&h);
#pragma unroll
              for (int e = 0; e < 8; e++) {
                const float x = (float)hb[e];
                v[e] = ACT == 5 ? gelu_f(x) : x * sigmoidf_(x);
              }
            }
            if constexpr (OUT_F32) {
              const f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
              keep[2 * q] = *reinterpret_cast<const u32x4*>(&o0);
              keep[2 * q + 1] = *reinterpret_cast<const u32x4*>(&o1);
              __builtin_amdgcn_raw_buffer_store_b128(keep[2 * q], rY, ok ? eo * 4u : SENT, 0, 0);
              __builtin_amdgcn_raw_buffer_store_b128(keep[2 * q + 1], rY, ok ? eo * 4u + 16u : SENT, 0, 0);
            } else {
              const u32x4 o = {dmvae_pack_bf16x2(v[0], v[1]), dmvae_pack_bf16x2(v[2], v[3]), dmvae_pack_bf16x2(v[4], v[5]), dmvae_pack_bf16x2(v[6], v[7])};
              keep[q] = o;
#if !(DMVAE_GEMM_EXP & 1)
              __builtin_amdgcn_raw_buffer_store_b128(o, rY, ok ? eo * 2u : SENT, 0, (DMVAE_GEMM_EXP & 4) ? 2 : 0);
#endif
            }
          }
          // gfx950: a VALU write to a store's data VGPR directly behind the store is seen by the store (conv_pp.hip, tools/probes/probe_store_war.hip) --
          // every packed result stays live up to here
#pragma unroll
          for (int q = 0; q < (OUT_F32 ? 2 * NI : NI); q++) asm volatile("s_nop 0" :: "v"(keep[q]));
          if constexpr (ACT != 0) {
#pragma unroll
            for (int q = 0; q < NI; q++) asm volatile("s_nop 0" :: "v"(keep2[q]));
          }
        }
      };
      if (a.act == 0) body(std::integral_constant<int, 0>{});
      else if (a.act == 5) body(std::integral_constant<int, 5>{});
      else body(std::integral_constant<int, 1>{});
    }
    if (has_next) issue(SLOT);
    wait_vmcnt<NP>();                // the stores share vmcnt with the prefetched K tiles: everything but the newest tile's pieces has landed
    __builtin_amdgcn_s_barrier();    // staging reads done before ring slots 2.. are refilled
    if (has_next) issue(2 * SLOT);
    work = next;
  }
#endif
}

template <int TM, int TP, int WM, int WP, bool F32>
int launch(Args a, hipStream_t st) {
  a.ntn = (a.N + TM - 1) / TM;
  a.total = ((a.M + TP - 1) / TP) * a.ntn;
  const unsigned grid = a.total > 256 ? 256u : (unsigned)a.total;
  constexpr int slot = (TM + TP) * 64, cw = TM / WM, epi = 2 * slot + 8 * 16 * (cw * 4 + 16);
  constexpr bool dump = ((TM / 16 + 7) / 8) * 128 > TM || ((TP / 16 + 7) / 8) * 128 > TP;   // some wave's piece lies past the tile: it lands in a dump KiB
  constexpr int lds = (4 * slot > epi ? 4 * slot : epi) + (dump ? 1024 : 0);
  static_assert(lds <= 160 * 1024, "LDS");
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_pp_kernel<TM, TP, WM, WP, F32>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done = true;
  }
  hipLaunchKernelGGL((gemm_pp_kernel<TM, TP, WM, WP, F32>), dim3(grid), dim3(512), lds, st, a);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

// The tile menu: (columns, rows, relative cost of one K step of the tile in units of a 256 x 256 tile's -- measured, tools/bench_gemm.py --sweep).
struct Cfg { int tm, tp; float cost; };
static const Cfg g_cfg[] = {
    {256, 256, 1.00f}, {256, 224, 0.90f}, {256, 192, 0.79f}, {256, 160, 0.68f}, {256, 128, 0.57f},
    {128, 512, 1.05f}, {128, 384, 0.80f}, {128, 256, 0.57f}, {192, 256, 0.80f}, {192, 320, 0.98f},
};
constexpr int NCFG = sizeof(g_cfg) / sizeof(g_cfg[0]);

static int g_forced = -2;   // -2: not read yet; -1: plan by cost; >= 0: this menu entry (DMVAE_GEMM_CFG, or dmvae_debug_gemm_cfg from tools/bench_gemm.py)
static int plan(int M, int N, int K) {
  if (g_forced == -2) { const char* e = getenv("DMVAE_GEMM_CFG"); g_forced = e ? atoi(e) : -1; }
  const int forced = g_forced;
  if (forced >= 0 && forced < NCFG) return forced;
  (void)K;
  int best = 0;
  float best_t = 1e30f;
  for (int c = 0; c < NCFG; c++) {
    const long long tiles = (long long)((M + g_cfg[c].tp - 1) / g_cfg[c].tp) * ((N + g_cfg[c].tm - 1) / g_cfg[c].tm);
    const long long rounds = (tiles + 255) / 256;
    // padding columns are wasted work inside a tile (N = 1152 on 256-column tiles), padding rows likewise: both are in `tiles`
    const float t = (float)rounds * g_cfg[c].cost;
    if (t < best_t * 0.999f) { best_t = t; best = c; }
  }
  return best;
}

template <bool F32>
static int dispatch(int cfg, const Args& a, hipStream_t st) {
  switch (cfg) {
    case 0: return launch<256, 256, 2, 4, F32>(a, st);
    case 1: return launch<256, 224, 4, 2, F32>(a, st);
    case 2: return launch<256, 192, 2, 4, F32>(a, st);
    case 3: return launch<256, 160, 4, 2, F32>(a, st);
    case 4: return launch<256, 128, 4, 2, F32>(a, st);
    case 5: return launch<128, 512, 2, 4, F32>(a, st);
    case 6: return launch<128, 384, 2, 4, F32>(a, st);
    case 7: return launch<128, 256, 2, 4, F32>(a, st);
    case 8: return launch<192, 256, 2, 4, F32>(a, st);
    default: return launch<192, 320, 2, 4, F32>(a, st);
  }
}

}  // namespace dmvae_gemm_pp

extern "C" void dmvae_debug_gemm_cfg(int cfg) { dmvae_gemm_pp::g_forced = cfg; }   // diagnostics only (tools/bench_gemm.py): force a menu entry, -1 = plan by cost

extern "C" int dmvae_linear_bf16_plan(int M, int N, int K, int* tile_cols, int* tile_rows) {
  using namespace dmvae_gemm_pp;
  const int c = plan(M, N, K);
  if (tile_cols) *tile_cols = g_cfg[c].tm;
  if (tile_rows) *tile_rows = g_cfg[c].tp;
  return c;
}

extern "C" int dmvae_linear_bf16(const void* x, const void* w, const void* bias, void* y, void* y_pre, int M, int N, int K, int lda, int ldw, int ldy,
                                 int act, int bias_bf16, int out_f32, hipStream_t stream) {
  using namespace dmvae_gemm_pp;
  DMVAE_CHECK_ARG(x && w && y, "linear_bf16: null operand");
  DMVAE_CHECK_ARG(M > 0 && N > 0 && K >= 32 && K % 32 == 0 && N % 8 == 0, "linear_bf16: need K %% 32 == 0 and N %% 8 == 0 (M %d, N %d, K %d)", M, N, K);
  DMVAE_CHECK_ARG(lda >= K && ldw >= K && ldy >= N && lda % 8 == 0 && ldw % 8 == 0 && ldy % 8 == 0, "linear_bf16: leading dimensions must cover the rows and be multiples of 8");
  DMVAE_CHECK_ARG(act == 0 || act == 1 || act == 5, "linear_bf16: act must be 0 (none), 1 (SiLU) or 5 (GELU)");
  DMVAE_CHECK_ARG(!y_pre || act != 0, "linear_bf16: y_pre (the bf16 pre-activation) needs an activation");
  DMVAE_CHECK_ARG((long long)M * lda * 2 < (1ll << 31) && (long long)N * ldw * 2 < (1ll << 31) && (long long)M * ldy * (out_f32 ? 4 : 2) < (1ll << 31),
                  "linear_bf16: operands are addressed through 32-bit buffer offsets (2 GiB each)");
  Args a;
  a.x = (const bf16*)x; a.w = (const bf16*)w; a.bias = bias; a.y = y; a.y2 = act != 0 ? (bf16*)y_pre : nullptr;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldw = ldw; a.ldy = ldy;
  a.act = act; a.bias_bf16 = bias_bf16; a.ntn = 0; a.total = 0;
  const int cfg = plan(M, N, K);
  return out_f32 ? dispatch<true>(cfg, a, stream) : dispatch<false>(cfg, a, stream);
}
