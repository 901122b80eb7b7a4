// Linear-layer GEMM  Y[M][N] = act(X[M][K] . W[N][K]^T + bias)  for bf16 operands on gfx950 -- the transformer blocks' nn.Linear under autocast(bf16):
// timm's ViT blocks reached through models/vae.py:47-53 (qkv / proj / fc1 / fc2), diffusion/lightningdit/lightningdit.py:34-93,173-252 (qkv / proj),
// swiglu_ffn.py:15-36 (w12 / w3), and -- on a transposed bf16 copy of the weight -- their input gradients dX = dY . W.
//
// The loop is conv_pp.hip's ping-pong loop without the convolution: 8 waves, both operands K-contiguous and staged by LDS-DMA in K tiles of 32 (64-B LDS rows,
// XOR-swizzled 16-B chunks, conflict-free ds_read_b128 fragments for v_mfma_f32_16x16x32_bf16), a 4-deep ring filled three K tiles ahead under a counted vmcnt,
// the two waves of a SIMD alternating LOAD and COMPUTE intervals, persistent blocks (one per CU) walking an XCD-aware tile order, an LDS-staged epilogue that
// writes whole output rows.  What is new is the TILE: these GEMMs have few tiles per CU (M = batch x tokens = 8224 for ViT-L at batch 32 is 32.125 tiles of
// 256 rows, N = 1024 is four tiles of 256 columns: 132 tiles for 256 CUs), so what a call costs is decided by tile quantisation -- the vendor library's picks
// measure exactly "rounds x one 256 x 256 tile" (DESIGN_HISTORY.md 8.12).  Here the tile height is a template parameter in steps of 32 rows (wave grid 4 x 2, each wave
// 64 columns x 16 * BP16 rows) or 64 rows (wave grid 2 x 4), and the host picks, per (M, N, K), the instantiation whose round count x tile cost is lowest
// (dmvae_gemm_pp_plan): e.g. proj (N = 1024) runs 208 tiles of 256 x 160 in ONE round instead of 132 of 256 x 256 on half the chip.
//
// Epilogue: bias (f32, or bf16 the way autocast hands it to the library) as the accumulators' initial value, optional exact GELU (nn.GELU() of timm's Mlp) or
// SiLU applied to the bf16-rounded pre-activation -- bit-identical to the Linear followed by csrc/vit_bwd.hip::gelu_fwd_kernel / misc.hip::silu_fwd_kernel --,
// bf16 or f32 result.  Rows past M / columns past N never leave the CU: every store's per-lane offset is out of the descriptor's range for them (the range
// check does not see the scalar offset, so validity never rides on it).
#include "common.h"
#include "dmvae_hip.h"
#include <cstdlib>
#include <type_traits>

// Measured on this kernel and not adopted (kept as text, DESIGN_HISTORY.md 9.1): a hand-pipelined single-stream main loop with one barrier per K tile (and with one every
// second K tile), a four-wave 128 x 128 wave tile, a deeper / shallower ring, nt stores in the epilogue.
constexpr int GEMM_MAXBUF = 6;   // ring depth cap; the ring is as deep as LDS allows up to this

namespace dmvae_gemm_pp {

struct Args {
  const bf16* x;     // [M][lda]
  const bf16* w;     // [N][ldw], or K-tile-major [K / 32][N][32]
  const void* bias;  // [N] f32 / bf16, or null
  void* y;           // [M][ldy] bf16 / f32
  int M, N, K, lda, ldw, ldy;
  int act, bias_bf16;
  int H;             // act 6 (SwiGLU): N = 2 H weight rows [x1 | x2], H output columns
  unsigned wbytes;       // bytes of the weight operand
  unsigned wsRow, wsK;   // byte strides of the weight operand between rows / between K tiles: row-major (2 ldw, 64), K-tile-major (64, 64 N)
  int ntn, total;    // column tiles, tiles
  float inv_ntn;     // 1 / ntn
  // BATCHED instantiations (dmvae_linear_bf16_batched: `batch` independent products y_b = x_b w_b^T, e.g. the decoder attention's q k^T per sample): tile index ->
  // (b, tile of one product); byte strides between the products' operands; the buffer descriptors span the whole batch
  int tpb;           // tiles per product
  float inv_tpb;
  unsigned sA, sB, sY, xbytes, ybytes;
  unsigned long long* dbg;   // optional s_memtime stamps per block (dmvae_debug_gemm_timing): [block][tile 0..3][4], null in production
  // SK instantiations (dmvae_linear_bf16_sk): the flattened (tile, K step) space -- T = tiles x K / 32 steps -- is cut into V contiguous ranges ("virtual
  // blocks"), each walked by one workgroup; a range that ends inside a tile leaves a PARTIAL tile, and the last of a tile's parts to arrive sums them in
  // K order (see the kernel).  skS > 0: V = tiles x skS uniform parts per tile (the cut depends on N and K only); skS = 0: stream-K, V equal ranges.
  int V, skS, T;
  float* slabs;              // [2 V][TM x TP] f32 partial tiles, each in its writer's register layout
  unsigned* counters;        // [tiles][8 waves] arrival counts; zero on entry, left zero
  unsigned slab_bytes;
  // act 6 only (dmvae_linear_bf16_swiglu_pre): the bf16-rounded PRE-activation [x1 | x2] is stored too ([M][ldy2], what the backward of SwiGLU reads) -- the
  // w12 Linear and swiglu_ffn.py:32-35's product in one launch, no separate pass over the 2 H-wide tensor
  void* y2;
  int ldy2;
};

// Cache policy of the SK instantiation's partial-tile traffic (aux bits of the buffer builtins on gfx940+: 1 = sc0, 16 = sc1): sc0 sc1 = system scope -- the store
// is written through this XCD's L2 to memory and the load misses in the reader's L2 --, so a part written on one XCD is read correctly on another WITHOUT
// whole-cache maintenance.  A first version used __threadfence() on both sides (buffer_wbl2 / buffer_inv, i.e. one L2 write-back + invalidate per wave and part):
// 1 900 cache-wide operations in a 30-us kernel serialised per XCD and made it 3-5 x slower than the unsplit form (profiles/r6_gemm_sk_first_fences.txt).
constexpr int SK_COHERENT = 1 | 16;
constexpr unsigned SENT = 0x80000000u;  // voffset beyond any descriptor's num_records: loads return zeros, stores are dropped

__device__ __forceinline__ int swz64(int row) { return (0 - (row >> 2)) & 3; }   // conv_pp.hip: the 64-B-row chunk key that keeps ds_read_b128 fragments conflict-free

template <int N>
__device__ __forceinline__ void wait_vmcnt() {   // through the builtin: the compiler's wait-count pass has to SEE the wait (conv_pp.hip)
  __builtin_amdgcn_s_waitcnt((N & 15) | 0x0F70 | ((N >> 4) << 14));
  asm volatile("" ::: "memory");
}

__device__ __forceinline__ float gelu_f(float x) { return dmvae_gelu_f(x); }   // common.h; vit_bwd.hip::gelu_fwd_kernel computes the same bits

// TM: output columns (weight rows) per tile, TP: output rows (tokens) per tile, WM x WP: wave grid over (columns, rows).
//
// One continuous K-tile stream per block: the ring does not care which output tile a K tile belongs to, so the last three K steps of a tile issue the first three
// K tiles of the block's NEXT tile (and its bias slice, one more 1-KiB piece into a two-slot LDS region) and the next tile's loop starts on operands that have
// already landed.
//
// Epilogue straight from the accumulators, no LDS: the MFMA runs with the TOKENS as its rows (D[token][column]: lane (c = lane & 15, g = lane >> 4) holds tokens
// 4g .. 4g + 3 of a token block at column c of a column block), and the weight rows are DMA'd in a permuted order -- LDS row 16 i + c of a wave's column range
// holds column CL * c + i (CL = column blocks per wave) -- so that a lane's CL accumulators of one token are CL CONSECUTIVE columns and the 16 lanes of a group
// hold 16 CL consecutive columns of that token: one store instruction writes 4 rows x 256 contiguous bytes (CL = 8) with consecutive lanes on consecutive bytes.
// History, measured with s_memtime stamps (tools/probes/time_gemm_pp.py, k-cycles per 256 x 192 tile): staging the tile through LDS in f32 like conv_pp 6-8;
// columns along the MFMA rows + a DPP exchange between lanes m and m ^ 8 (8 rows x 128 B per instruction, but a row's chunks on lanes 8 apart) 6.5, of which
// stores 4.0 (42 cycles per store instruction, the same with or without the exchange: what the store path wants is consecutive lanes on consecutive bytes).
// The bias is the accumulators' initial value (read from the LDS slice in the MFMA layout), not an epilogue add.
//
// vmcnt bookkeeping (per wave; P = NP pieces per K tile).  LDS-DMA loads retire in order among themselves, but stores do NOT retire in order with loads (a first
// version that counted on it -- "[K1'][K2'][S stores][K3]: at most 2P + S outstanding means K1' has landed" -- read K tiles before they had landed whenever the
// stores were acknowledged first: wrong results on the first call, right ones on a rerun that found the same bytes still in LDS).  So no wait ever leaves a store
// AND a needed load behind it: a count is only trusted where it forces every store to have retired.
//   kernel start        [bias K0][K1] .. [K(PF-1)]                       wait 0          (PF = NBUF - 1 K tiles ahead: as deep as 160 KB of LDS allow for the tile --
//                                                                                        three K steps, 1.5 us, do not cover an HBM miss: with inputs the previous
//                                                                                        kernel wrote instead of inputs the same GEMM just read, the 4-slot ring of the
//                                                                                        256 x 160 tile measured 81 us on fc2 against 67 warm; tools/bench_gemm.py --cold)
//   tile entry          K0 .. K(PF-1) and the bias slice have landed; the previous tile's stores may be in flight
//   t = 0 .. PF - 2     + K(PF + t)                                      no wait (K(t + 1) landed at the previous tile's end)
//   t = PF - 1          [S stores in any state][K(PF)] .. [K(2PF - 1)]   wait (PF-1) P: at most that many outstanding = every store and K(PF) retired
//   t = PF .. nK-PF-1   steady state                                     wait (PF-1) P
//   t = nK-PF .. nK-2   + [bias' K0'] (P + 1 pieces), + K1' ..           wait (PF-1) P + 1
//   t = nK - 1          + K(PF-1)'                                       wait 0: the next tile's first PF K tiles have landed before this tile's stores go out
// Needs nK >= 2 PF (the host asks for K >= 384 and sends shorter reductions to the small batched kernel).  No scratch: a spilled register's reload is a VMEM load, and
// the wait the compiler puts behind it drains the whole prefetch queue (measured: 10 k cycles per tile with 31 spilled VGPRs).
//
// SK (stream-K / fused split-K; round 6).  Few-tile deep-K problems leave CUs idle (M 4096 x N 1152 is 80 tiles of 256 x 256 for 256 CUs) and problems of
// 1.x rounds waste most of the last round.  Here a workgroup walks a contiguous RANGE of the flattened (tile, K step) space instead of whole tiles: a unit of
// work is (tile, first K step kb, nk steps).  The loop below is unchanged -- it already streams K tiles across tile boundaries; a unit's K offset rides on the
// scalar offset of its loads.  A unit with nk < K / 32 is a PART of its tile: its accumulators go to a slab slot in the wave's own register layout (1 KiB per
// store instruction, written through to memory), then -- once the stores are acknowledged -- one atomic per wave on the tile's arrival counter; the wave that
// arrives LAST sums the parts in K
// order -- its own from the accumulators, the others from their slots: (((bias + p0) + p1) + p2 ...), the same order whoever arrives last, so results are
// run-to-run identical -- and runs the ordinary epilogue.  Nobody ever waits for another workgroup (no co-residency assumption, no spinning).  Cuts: uniform
// (skS parts per tile; depends on N, K only -- an output row's bits do not depend on M, which the DMD loss's batched cond / uncond evaluation relies on) or
// stream-K (V = grid equal ranges, each cut kept MINSEG steps away from tile edges).
template <int TM, int TP, int WM, int WP, bool OUT_F32, int NBUF, bool BATCHED = false, bool SK = false>
__global__ __launch_bounds__(WM * WP * 64) void gemm_pp_kernel(Args a) {
#if __HIP_DEVICE_COMPILE__
  constexpr int PF = NBUF - 1;   // K tiles in flight ahead of the one being read
  static_assert(!(SK && BATCHED) && !(SK && OUT_F32), "SK: single product, bf16 result");
  constexpr int CL = TM / WM / 16, BT = TP / WP / 16;   // 16 x 16 accumulator blocks per wave: CL column blocks (= consecutive columns per lane) x BT token blocks
  constexpr int NW = WM * WP;   // waves: 8 (two per SIMD), or 4 with the single-stream loop (one per SIMD, up to 64 accumulator blocks each: a 128 x 128 wave tile)
  static_assert(NW == 8 && CL * 16 * WM == TM && BT * 16 * WP == TP && CL * BT <= 32 && (CL == 4 || CL == 6 || CL == 8), "wave grid");
  constexpr int TILE_A = TM * 64, TILE_B = TP * 64, SLOT = TILE_A + TILE_B, RING = NBUF * SLOT;
  constexpr int NPA = (TM / 16 + NW - 1) / NW, NPB = (TP / 16 + NW - 1) / NW, NP = NPA + NPB;   // 1-KiB DMA pieces (16 rows x 64 B) per wave and K tile; pieces past the tile go to the dump KiB
  constexpr int CW = CL * 16;                      // output columns per wave
  constexpr int BIAS_OFF = RING, DUMP_OFF = RING + 2048;
  constexpr int SPR = OUT_F32 ? (CL + 3) / 4 : 1;  // store instructions per lane and token row
  constexpr int NST = BT * 4 * SPR;                // ... per lane and tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const int wm = wave / WP, wp = wave % WP;
  const int nK = a.K >> 5;
  const unsigned bsz = a.bias_bf16 ? 2u : 4u;
  const bool gated = a.act == 6;
  const float inv_ntn = a.inv_ntn;   // from the host: a kernel argument lives in an SGPR (computed here it sat in a VGPR and was spilled)

  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.wbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, BATCHED ? a.xbytes : (unsigned)a.M * (unsigned)a.lda * 2u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rBias = __builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0, a.bias ? (unsigned)a.N * bsz : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, BATCHED ? a.ybytes : (unsigned)a.M * (unsigned)a.ldy * (OUT_F32 ? 4u : 2u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rY2 = __builtin_amdgcn_make_buffer_rsrc(a.y2, 0, a.y2 ? (unsigned)a.M * (unsigned)a.ldy2 * 2u : 0u, 0x00020000);

  // per-lane DMA source offsets of a tile: weight rows in the permuted order, token rows as they are, the bias slice (wave 0; 16 B per lane)
  auto calc = [&](unsigned work, unsigned (&vA)[NPA], unsigned (&vB)[NPB], unsigned& vBias, int& m0, int& n0, int& b0) {
    int lane_c = lane;
    asm volatile("" : "+v"(lane_c));   // an opaque copy: what calc derives from the lane index is recomputed per tile, not hoisted out of the tile loop and carried (spilled) across the K loop
    const bool live = work < (unsigned)a.total;
    int wid = live ? (SK ? (int)work : (int)xcd_remap(work, a.total)) : 0;   // SK: `work` is the unit's tile (the XCD-aware order is applied to the virtual blocks)
    unsigned offA = 0, offB = 0;
    b0 = 0;
    if constexpr (BATCHED) {   // product b, its tile wid - b tpb (the same reciprocal + fix-up as below)
      int b = (int)((float)wid * a.inv_tpb);
      int r = wid - b * a.tpb;
      if (r < 0) { b--; r += a.tpb; }
      if (r >= a.tpb) { b++; r -= a.tpb; }
      b0 = __builtin_amdgcn_readfirstlane(b);
      wid = r;
      offA = (unsigned)b0 * a.sA; offB = (unsigned)b0 * a.sB;
    }
    int mt = (int)((float)wid * inv_ntn);            // wid / ntn for wid < 2^24 (the host keeps the tile count below that): one reciprocal and a +-1 fix-up
    int nt = wid - mt * a.ntn;
    if (nt < 0) { mt--; nt += a.ntn; }
    if (nt >= a.ntn) { mt++; nt -= a.ntn; }
    m0 = __builtin_amdgcn_readfirstlane(mt) * TP;
    n0 = __builtin_amdgcn_readfirstlane(nt) * TM;
    // Every tile starts its reduction at K tile 0 and the CUs walk K in step: the tiles of one row / column of tiles, running side by side, ask for the same
    // operand lines at the same time and all but one request hit in L2.  (A per-tile starting K tile, to spread the chip's requests over more addresses, was
    // measured and destroys exactly that: fc2 70 -> 98 us.)
#pragma unroll
    for (int p = 0; p < NPA; p++) {
      const int rl = (wave * NPA + p) * 16 + (lane_c >> 2);     // LDS row of the weight tile: wave column range rl / CW, column block i, MFMA column c
      const int wmr = rl / CW, r = rl % CW, i = r >> 4, c = r & 15;
      int n = n0 + wmr * CW + CL * c + i;
      bool n_ok = n < a.N;
      if (gated) {   // a lane's CL accumulators: x1 of CL / 2 consecutive hidden units, then x2 of the same units (weight rows H + unit)
        const int hid = (n0 >> 1) + wmr * (CW / 2) + (CL / 2) * c + (i % (CL / 2));
        n = (i < CL / 2 ? 0 : a.H) + hid;
        n_ok = hid < a.H;
      }
      const int ch = (lane_c & 3) ^ swz64(rl);                  // logical 16-B chunk this lane fetches: the LDS image stays lane-linear, the swizzle is on the source address
      vA[p] = (live && rl < TM && n_ok) ? (unsigned)n * a.wsRow + ch * 16u + offA : SENT;
    }
#pragma unroll
    for (int p = 0; p < NPB; p++) {
      const int rl = (wave * NPB + p) * 16 + (lane_c >> 2);
      const int m = m0 + rl;
      const int ch = (lane_c & 3) ^ swz64(rl);
      vB[p] = (live && rl < TP && m < a.M) ? (unsigned)m * (unsigned)a.lda * 2u + ch * 16u + offB : SENT;
    }
    const unsigned e0 = (unsigned)lane_c * (16u / bsz);         // first element of this lane's 16 bytes of the slice
    if (gated) {   // slice = [x1 biases of the tile's TM / 2 hidden units][x2 biases of the same units]
      const unsigned eh = e0 < (unsigned)(TM / 2) ? e0 : e0 - (unsigned)(TM / 2);
      const unsigned hid = (unsigned)(n0 >> 1) + eh;
      vBias = (live && wave == 0 && e0 < (unsigned)TM && hid < (unsigned)a.H) ? ((e0 < (unsigned)(TM / 2) ? 0u : (unsigned)a.H) + hid) * bsz : SENT;
    } else {
      vBias = (live && wave == 0 && e0 < (unsigned)TM && (unsigned)n0 + e0 < (unsigned)a.N) ? ((unsigned)n0 + e0) * bsz : SENT;
    }
  };
  auto issue_k = [&](const unsigned (&vA)[NPA], const unsigned (&vB)[NPB], unsigned kt, int slot) {   // kt: K tile
#pragma unroll
    for (int p = 0; p < NPA; p++) {
      const int g = wave * NPA + p;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, LPTR(smem + (g * 16 < TM ? slot + g * 1024 : DUMP_OFF)), 16, vA[p], kt * a.wsK, 0, 0);
    }
#pragma unroll
    for (int p = 0; p < NPB; p++) {
      const int g = wave * NPB + p;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, LPTR(smem + (g * 16 < TP ? slot + TILE_A + g * 1024 : DUMP_OFF)), 16, vB[p], kt * 64u, 0, 0);
    }
  };
  auto issue_bias = [&](unsigned vBias, int par) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rBias, LPTR(smem + (wave == 0 ? BIAS_OFF + par * 1024 : DUMP_OFF)), 16, vBias, 0, 0, 0);
  };

  // fragment reads: lane l takes the 16-B chunk l >> 4 of row l & 15 of a 16-row block; the swizzle key of row 16 b + y is that of row y (swz64 sees y >> 2 mod 4 only),
  // so every block of a tile is the first block's address + b * 1024 -- one address register per operand and immediate offsets
  const int fr = (lane & 15) * 64 + (((lane >> 4) ^ swz64(lane & 15)) << 4);
  const int wbase = wm * (CW * 64) + fr;                       // weight tile: rows wm * CW ..
  const int tbase = TILE_A + wp * (TP / WP) * 64 + fr;         // token tile:  rows wp * (TP / WP) ..

  // ---- SK: units of work ------------------------------------------------------------------------------------------------------------------------------
  constexpr int MINSEG = 2 * PF + 2;   // shortest part of a tile (K steps): the loop below needs >= 2 PF, two more keep a part from being all prologue
  struct Unit { int tile, kb, nk, p, P, i1; };   // tile; first K step and K steps of this unit; part index / parts of the tile; index of the tile's first interior cut
  auto bnd = [&](int v) -> int {       // flattened step at which virtual block v begins (v = V: the end)
    if (a.skS > 0) {
      const int t = v / a.skS, p = v - t * a.skS;
      return t * nK + (p * nK) / a.skS;
    }
    int b = (int)((long long)v * (long long)a.T / (long long)a.V);
    const int r = b % nK;
    if (r < MINSEG) b -= r;
    else if (nK - r < MINSEG) b += nK - r;
    return b;
  };
  unsigned sk_work = blockIdx.x;   // position in the list of virtual blocks this workgroup walks: blockIdx.x, + gridDim.x, ...
  int sk_pos = 0, sk_end = 0, sk_v = 0;
  bool sk_started = false;
  auto next_unit = [&]() -> Unit {
    Unit u = {a.total, 0, 2 * PF, 0, 1, 0};          // tile = total: not live (its pieces move no memory; the loop shape stays valid)
    while (sk_pos >= sk_end) {
      if (sk_started) sk_work += gridDim.x;
      sk_started = true;
      if (sk_work >= (unsigned)a.V) return u;
      sk_v = (int)xcd_remap(sk_work, (unsigned)a.V);
      sk_pos = bnd(sk_v); sk_end = bnd(sk_v + 1);
    }
    const int tile = sk_pos / nK, lo = tile * nK, hi = lo + nK;
    u.tile = tile; u.kb = sk_pos - lo;
    const int ke = sk_end < hi ? sk_end - lo : nK;
    u.nk = ke - u.kb;
    sk_pos += u.nk;
    if (u.nk != nK) {                                // a part: the tile's interior cuts are bnd(i1) .. bnd(w - 1), consecutive virtual blocks
      int v = sk_v;
      while (v >= 1 && bnd(v) > lo) v--;
      u.i1 = v + 1;
      int w = sk_v + 1;
      while (w < a.V && bnd(w) < hi) w++;
      u.P = w - u.i1 + 1;
      u.p = u.kb == 0 ? 0 : sk_v - u.i1 + 1;
    }
    u.tile = __builtin_amdgcn_readfirstlane(u.tile); u.kb = __builtin_amdgcn_readfirstlane(u.kb); u.nk = __builtin_amdgcn_readfirstlane(u.nk);
    u.p = __builtin_amdgcn_readfirstlane(u.p); u.P = __builtin_amdgcn_readfirstlane(u.P); u.i1 = __builtin_amdgcn_readfirstlane(u.i1);
    return u;
  };
  Unit uc = {0, 0, nK, 0, 1, 0}, un = uc;            // current / next unit (non-SK: every unit is a whole tile)

  f32x4 acc[CL][BT];   // acc[i][j][r]: token 16 j + 4 (lane >> 4) + r of the wave's rows; column CL * (lane & 15) + i of the wave's columns
  bf16x8 wf[CL], tf[BT];
  unsigned vAc[NPA], vBc[NPB], vAn[NPA], vBn[NPB], vBiasN;
  int m0c, n0c, m0n = 0, n0n = 0, b0c = 0, b0n = 0;
  int slot_rd = 0, slot_wr = PF * SLOT, par = 0;
  int it = PF;             // next K step of the current output tile to issue


  // one K step: LOAD interval (fragments of the slot being read, DMA issue into the slot being written), barrier, COMPUTE interval, barrier
  auto kstep = [&](auto WAITc, auto MODEc) __attribute__((always_inline)) {
    constexpr int WAIT = decltype(WAITc)::value, MODE = decltype(MODEc)::value;   // MODE 0: the current tile's K tile `it`; j >= 1: the next tile's K tile j - 1 (+ the bias slice with j = 1)
    const char* sw = smem + slot_rd + wbase;
    const char* st = smem + slot_rd + tbase;
#pragma unroll
    for (int j = 0; j < BT; j++) tf[j] = *reinterpret_cast<const bf16x8*>(st + j * 1024);
#pragma unroll
    for (int i = 0; i < CL; i++) wf[i] = *reinterpret_cast<const bf16x8*>(sw + i * 1024);
    if constexpr (MODE == 0) { issue_k(vAc, vBc, (unsigned)(it + (SK ? uc.kb : 0)), slot_wr); it++; }
    else {
      if constexpr (MODE == 1) issue_bias(vBiasN, par ^ 1);
      issue_k(vAn, vBn, (unsigned)(MODE - 1 + (SK ? un.kb : 0)), slot_wr);
    }
    slot_rd = slot_rd + SLOT == RING ? 0 : slot_rd + SLOT;
    slot_wr = slot_wr + SLOT == RING ? 0 : slot_wr + SLOT;
    if constexpr (WAIT >= 0) wait_vmcnt<WAIT>();   // own pieces of the NEXT K tile have landed
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0) through the builtin: the compiler sees the fragments have arrived and puts no waits of its own between the MFMAs
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < CL; i++)
#pragma unroll
      for (int j = 0; j < BT; j++)   // rows = tokens (first operand), columns = weight rows (second operand)
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(tf[j]), "v"(wf[i]));
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  using W_NO = std::integral_constant<int, -1>; using W_0 = std::integral_constant<int, 0>;
  using W_ST = std::integral_constant<int, (PF - 1) * NP>; using W_TL = std::integral_constant<int, (PF - 1) * NP + 1>;
  using M0_ = std::integral_constant<int, 0>;
  static_assert((PF - 1) * NP + 1 < 64, "vmcnt is a 6-bit counter");

  unsigned work = blockIdx.x;
  if constexpr (SK) { uc = next_unit(); work = (unsigned)uc.tile; }
  {
    unsigned vBiasC;
    calc(work, vAc, vBc, vBiasC, m0c, n0c, b0c);
    issue_bias(vBiasC, 0);
  }
#pragma unroll
  for (int u = 0; u < PF; u++) issue_k(vAc, vBc, (unsigned)(u + (SK ? uc.kb : 0)), u * SLOT);
  wait_vmcnt<0>();

  int tix = 0;
  auto stamp = [&](int k) { if (a.dbg && tid == 0 && tix < 4) a.dbg[((size_t)blockIdx.x * 4 + tix) * 8 + k] = __builtin_amdgcn_s_memtime(); };   // [block][tile 0..3][8]: 0-3 the tile's phases, 4-7 inside SK's hand-over
  while (work < (unsigned)a.total) {
    stamp(0);
    __builtin_amdgcn_s_barrier();                  // B_0: everybody's pieces of K tiles 0 .. 2 and of the bias slice have landed (each wave waited for its own)
    {
      // accumulators start at the bias: the lane's CL consecutive columns, the same for every token
      const char* bl = smem + BIAS_OFF + par * 1024;
      int lane_b = lane;
      asm volatile("" : "+v"(lane_b));   // recomputed per tile (see calc)
      const int col = gated ? wm * (CW / 2) + (CL / 2) * (lane_b & 15) : wm * CW + CL * (lane_b & 15);
      const int hop = gated ? TM / 2 - CL / 2 : 0;   // gated: the second half of the lane's columns sits TM / 2 entries further (the x2 half of the slice)
      float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (!a.bias || (SK && uc.p != 0)) {   // no bias: the slice's piece was all out of range (whether such a piece writes zeros or nothing is not relied upon); SK: the bias starts part 0 only
      } else if (a.bias_bf16) {
#pragma unroll
        for (int h = 0; h < CL / 2; h++) {
          const bf16x2 b2 = *reinterpret_cast<const bf16x2*>(bl + (col + (h >= CL / 4 ? hop : 0)) * 2 + h * 4);
          bv[2 * h] = (float)b2[0]; bv[2 * h + 1] = (float)b2[1];
        }
      } else {
#pragma unroll
        for (int h = 0; h < CL / 2; h++) {
          const f32x2 b2 = *reinterpret_cast<const f32x2*>(bl + (col + (h >= CL / 4 ? hop : 0)) * 4 + h * 8);
          bv[2 * h] = b2[0]; bv[2 * h + 1] = b2[1];
        }
      }
#pragma unroll
      for (int i = 0; i < CL; i++)
#pragma unroll
        for (int j = 0; j < BT; j++) acc[i][j] = f32x4{bv[i], bv[i], bv[i], bv[i]};
    }
    stamp(1);
    unsigned next = work + gridDim.x;
    const int nkc = SK ? uc.nk : nK;               // K steps of this unit
    if (grp == 1) __builtin_amdgcn_s_setprio(1);   // static priority for the second-dispatched half (MI355X_MICROARCH.md, two waves per SIMD, item 4)
    if (grp == 1) __builtin_amdgcn_s_barrier();    // group 1 runs one interval behind group 0
    it = PF;
    [&]<int... T>(std::integer_sequence<int, T...>) __attribute__((always_inline)) { (((void)T, kstep(W_NO{}, M0_{})), ...); }(std::make_integer_sequence<int, PF - 1>{});
    kstep(W_ST{}, M0_{});
#pragma unroll 1
    for (int t = PF; t < nkc - PF; t++) kstep(W_ST{}, M0_{});
    if constexpr (SK) { un = next_unit(); next = (unsigned)un.tile; }
    calc(next, vAn, vBn, vBiasN, m0n, n0n, b0n);       // all out of range when there is no next tile: its pieces move no memory
    [&]<int... J>(std::integer_sequence<int, J...>) __attribute__((always_inline)) { (kstep(W_TL{}, std::integral_constant<int, J + 1>{}), ...); }(std::make_integer_sequence<int, PF - 1>{});
    kstep(W_0{}, std::integral_constant<int, PF>{});
    if (grp == 0) __builtin_amdgcn_s_barrier();    // matches group 1's extra barrier
    if (grp == 1) __builtin_amdgcn_s_setprio(0);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // MFMA (inline asm, invisible to the hazard recogniser) -> accumulator reads
    stamp(2);

    // ---- SK: a PART of a tile -> its slab slot; the wave that arrives last at the tile's counter sums the parts (K order) and goes on to the epilogue ---------------
    bool sk_final = true;                          // this wave writes the tile's result (always, outside SK)
    unsigned sk_base = 0;                          // byte offset of this wave's region inside a slab slot
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc((void*)a.slabs, 0, SK ? a.slab_bytes : 0u, 0x00020000);
    constexpr unsigned SLOT_BYTES = (unsigned)TM * (unsigned)TP * 4u, WREG = (unsigned)CL * BT * 4u * 64u * 4u;   // one partial tile; one wave's CL x BT x 4 floats per lane
    if constexpr (SK) {
      if (uc.P > 1) {
        static_assert(!SK || CL % 4 == 0, "SK: 16-B pieces of a lane's columns");
        sk_base = (unsigned)wave * WREG;
        const unsigned own = (uc.p == 0 ? (unsigned)(a.V + uc.i1) : (unsigned)(uc.i1 + uc.p - 1)) * SLOT_BYTES + sk_base;
        int lane_s = lane;
        asm volatile("" : "+v"(lane_s));
        const unsigned vo = (unsigned)lane_s * 16u;
#pragma unroll
        for (int j = 0; j < BT; j++)
#pragma unroll
          for (int r = 0; r < 4; r++) {
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < CL; i++) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v[i]) : "a"(acc[i][j][r]));
#pragma unroll
            for (int h = 0; h < CL / 4; h++) {
              const f32x4 o4 = {v[4 * h], v[4 * h + 1], v[4 * h + 2], v[4 * h + 3]};
              const u32x4 k4 = *reinterpret_cast<const u32x4*>(&o4);
              __builtin_amdgcn_raw_buffer_store_b128(k4, rS, vo, own + (unsigned)(((j * 4 + r) * (CL / 4) + h) * 1024), SK_COHERENT);
              asm volatile("s_nop 0" :: "v"(k4));
            }
          }
        stamp(4);
        wait_vmcnt<0>();                           // the part has been written THROUGH to memory (sc0 sc1 stores are acknowledged from there) before the arrival is counted
        stamp(5);
        unsigned old = 0;
        unsigned* cnt = a.counters + (size_t)uc.tile * 8 + wave;
        if (lane_s == 0) old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        old = (unsigned)__builtin_amdgcn_readfirstlane((int)old);
        sk_final = old == (unsigned)(uc.P - 1);
        stamp(6);
        if (sk_final) {
          if (lane_s == 0) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // left zero for the next launch
        }
      }
    }

    // ---- epilogue: straight from the accumulators, NST stores per lane: token row (j, r) of the lane's group, CL consecutive columns ------------------------------
    if (sk_final) {
      int lane_o = lane;
      asm volatile("" : "+v"(lane_o));   // recomputed per tile, not carried across the main loop
      const int col = gated ? (n0c >> 1) + wm * (CW / 2) + (CL / 2) * (lane_o & 15) : n0c + wm * CW + CL * (lane_o & 15);
      const int row0 = m0c + wp * (TP / WP) + 4 * (lane_o >> 4);
      const unsigned ysoff = BATCHED ? (unsigned)b0c * a.sY : 0u;   // the product's y: scalar offset of every store below (outside the descriptor's range check, like the K-tile offsets of the loads)
      const bool c_ok = col < (gated ? a.H : a.N);       // N % 8 == 0 and CL | 8 ... the lane's columns are all inside or all outside when N is a multiple of CL; else per element below
      auto body = [&](auto ACTc, auto SUMc) __attribute__((always_inline)) {
        constexpr int ACT = decltype(ACTc)::value;
        constexpr bool SUM = decltype(SUMc)::value;   // SK, last arriver of a tile in parts: a row's values are the K-ordered sum of the parts
#pragma unroll
        for (int j = 0; j < BT; j++) {
          [[maybe_unused]] float sm[4][8];
          if constexpr (SUM) {
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
              for (int i = 0; i < 8; i++) sm[r][i] = 0.f;     // 0 + p0 is p0 exactly: the order is ((p0 + p1) + p2) ... whoever sums
            const unsigned vo = (unsigned)lane_o * 16u;
#pragma unroll 1
            for (int q = 0; q < uc.P; q++) {
              if (q == uc.p) {
#pragma unroll
                for (int r = 0; r < 4; r++)
#pragma unroll
                  for (int i = 0; i < CL; i++) {
                    float t;
                    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t) : "a"(acc[i][j][r]));
                    sm[r][i] += t;
                  }
              } else {
                const unsigned src = (q == 0 ? (unsigned)(a.V + uc.i1) : (unsigned)(uc.i1 + q - 1)) * SLOT_BYTES + sk_base;
                u32x4 ld[4][CL / 4];
#pragma unroll
                for (int r = 0; r < 4; r++)
#pragma unroll
                  for (int h = 0; h < CL / 4; h++)
                    ld[r][h] = __builtin_amdgcn_raw_buffer_load_b128(rS, vo, src + (unsigned)(((j * 4 + r) * (CL / 4) + h) * 1024), SK_COHERENT);
#pragma unroll
                for (int r = 0; r < 4; r++)
#pragma unroll
                  for (int h = 0; h < CL / 4; h++) {
                    const f32x4 f = *reinterpret_cast<const f32x4*>(&ld[r][h]);
#pragma unroll
                    for (int e = 0; e < 4; e++) sm[r][4 * h + e] += f[e];
                  }
              }
            }
          }
#pragma unroll
          for (int r = 0; r < 4; r++) {
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if constexpr (SUM) {
#pragma unroll
              for (int i = 0; i < CL; i++) v[i] = sm[r][i];
            } else {
#pragma unroll
              for (int i = 0; i < CL; i++)   // pinned where they are used: left to the scheduler, the reads of later rows are hoisted and their values spilled (conv_pp.hip)
                asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v[i]) : "a"(acc[i][j][r]));
            }
            if constexpr (ACT == 6) {   // SwiGLU (swiglu_ffn.py:32-35): silu(x1) * x2 on the bf16-rounded halves, the product of bf16 values -- dit.hip::swiglu_kernel's bits
              if constexpr (!OUT_F32 && !BATCHED && (CL == 8 || CL == 4)) {
                if (a.y2) {   // the pre-activation as the unfused Linear would have stored it: x1 of the lane's CL / 2 hidden units at column hid, x2 at H + hid
                  const int m2 = row0 + j * 16 + r;
                  const bool ok2 = c_ok && m2 < a.M;
                  const unsigned e1 = ((unsigned)m2 * (unsigned)a.ldy2 + (unsigned)col) * 2u, e2 = e1 + (unsigned)a.H * 2u;
                  if constexpr (CL == 8) {
                    const u32x2 p1 = {dmvae_pack_bf16x2(v[0], v[1]), dmvae_pack_bf16x2(v[2], v[3])}, p2 = {dmvae_pack_bf16x2(v[4], v[5]), dmvae_pack_bf16x2(v[6], v[7])};
                    __builtin_amdgcn_raw_buffer_store_b64(p1, rY2, ok2 ? e1 : SENT, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(p2, rY2, ok2 ? e2 : SENT, 0, 0);
                    asm volatile("s_nop 0" :: "v"(p1), "v"(p2));
                  } else {
                    const unsigned p1 = dmvae_pack_bf16x2(v[0], v[1]), p2 = dmvae_pack_bf16x2(v[2], v[3]);
                    __builtin_amdgcn_raw_buffer_store_b32(p1, rY2, ok2 ? e1 : SENT, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(p2, rY2, ok2 ? e2 : SENT, 0, 0);
                    asm volatile("s_nop 0" :: "v"(p1), "v"(p2));
                  }
                }
              }
#pragma unroll
              for (int i = 0; i < CL / 2; i++) {
                const float x1 = (float)(bf16)v[i], x2 = (float)(bf16)v[CL / 2 + i];
                v[i] = (float)(bf16)(x1 * sigmoidf_(x1)) * x2;
              }
            } else if constexpr (ACT != 0) {   // the activation sees the bf16-rounded pre-activation, as the unfused Linear -> activation pair does
#pragma unroll
              for (int i = 0; i < CL; i++) {
                const float x = (float)(bf16)v[i];
                v[i] = ACT == 5 ? gelu_f(x) : x * sigmoidf_(x);
              }
            }
            const int m = row0 + j * 16 + r;
            const bool ok = c_ok && m < a.M;
            const unsigned eo = (unsigned)m * (unsigned)a.ldy + (unsigned)col;
            if constexpr (ACT == 6) {
              if constexpr (CL == 8 && !OUT_F32) {
                const u32x2 o = {dmvae_pack_bf16x2(v[0], v[1]), dmvae_pack_bf16x2(v[2], v[3])};
                __builtin_amdgcn_raw_buffer_store_b64(o, rY, ok ? eo * 2u : SENT, ysoff, 0);
                asm volatile("s_nop 0" :: "v"(o));
              } else if constexpr (CL == 4 && !OUT_F32) {
                const unsigned o = dmvae_pack_bf16x2(v[0], v[1]);
                __builtin_amdgcn_raw_buffer_store_b32(o, rY, ok ? eo * 2u : SENT, ysoff, 0);
                asm volatile("s_nop 0" :: "v"(o));
              }   // other instantiations are never dispatched with act 6
            } else if constexpr (OUT_F32 && CL == 6) {   // column pairs one by one (a lane may straddle N, see the bf16 case below; the f32 result is not a hot path)
#pragma unroll
              for (int e = 0; e < 3; e++) {
                const f32x2 o2 = {v[2 * e], v[2 * e + 1]};
                const u32x2 k2 = *reinterpret_cast<const u32x2*>(&o2);
                __builtin_amdgcn_raw_buffer_store_b64(k2, rY, (ok && col + 2 * e < a.N) ? eo * 4u + 8u * e : SENT, ysoff, 0);
                asm volatile("s_nop 0" :: "v"(k2));   // gfx950: a VALU write to a store's data VGPR directly behind the store is seen by the store (conv_pp.hip)
              }
            } else if constexpr (OUT_F32) {
              const f32x4 o0 = {v[0], v[1], v[2], v[3]};
              const u32x4 k0 = *reinterpret_cast<const u32x4*>(&o0);
              __builtin_amdgcn_raw_buffer_store_b128(k0, rY, ok ? eo * 4u : SENT, ysoff, 0);
              if constexpr (CL == 8) {
                const f32x4 o1 = {v[4], v[5], v[6], v[7]};
                const u32x4 k1 = *reinterpret_cast<const u32x4*>(&o1);
                __builtin_amdgcn_raw_buffer_store_b128(k1, rY, ok ? eo * 4u + 16u : SENT, ysoff, 0);
                asm volatile("s_nop 0" :: "v"(k0), "v"(k1));
              } else {
                asm volatile("s_nop 0" :: "v"(k0));
              }
            } else if constexpr (CL == 8) {
              const u32x4 o = {dmvae_pack_bf16x2(v[0], v[1]), dmvae_pack_bf16x2(v[2], v[3]), dmvae_pack_bf16x2(v[4], v[5]), dmvae_pack_bf16x2(v[6], v[7])};
              __builtin_amdgcn_raw_buffer_store_b128(o, rY, ok ? eo * 2u : SENT, ysoff, 0);
              asm volatile("s_nop 0" :: "v"(o));
            } else if constexpr (CL == 6) {
              // six columns per lane: N (a multiple of 8) need not be a multiple of 6, so one lane per row of the tile on N's edge straddles it -- that lane
              // stores its column pairs one by one (a wave-uniform branch that only the edge tiles take)
              const u32x3 o = {dmvae_pack_bf16x2(v[0], v[1]), dmvae_pack_bf16x2(v[2], v[3]), dmvae_pack_bf16x2(v[4], v[5])};
              const bool full = col + 6 <= a.N;
              __builtin_amdgcn_raw_buffer_store_b96(o, rY, (ok && full) ? eo * 2u : SENT, ysoff, 0);
              if (__builtin_amdgcn_ballot_w64(ok && !full) != 0ull) {
#pragma unroll
                for (int e = 0; e < 2; e++)
                  __builtin_amdgcn_raw_buffer_store_b32(o[e], rY, (ok && !full && col + 2 * e < a.N) ? eo * 2u + 4u * e : SENT, ysoff, 0);
              }
              asm volatile("s_nop 0" :: "v"(o));
            } else {
              const u32x2 o = {dmvae_pack_bf16x2(v[0], v[1]), dmvae_pack_bf16x2(v[2], v[3])};
              __builtin_amdgcn_raw_buffer_store_b64(o, rY, ok ? eo * 2u : SENT, ysoff, 0);
              asm volatile("s_nop 0" :: "v"(o));
            }
          }
        }
      };
      auto by_act = [&](auto SUMc) __attribute__((always_inline)) {
        if (a.act == 0) body(std::integral_constant<int, 0>{}, SUMc);
        else if (a.act == 5) body(std::integral_constant<int, 5>{}, SUMc);
        else if (a.act == 6) body(std::integral_constant<int, 6>{}, SUMc);
        else body(std::integral_constant<int, 1>{}, SUMc);
      };
      if constexpr (SK) {
        if (uc.P > 1) by_act(std::true_type{});
        else by_act(std::false_type{});
      } else {
        by_act(std::false_type{});
      }
    }
    if (a.dbg) __builtin_amdgcn_s_barrier();   // diagnostics: the stamp then reads when the LAST wave has issued its stores
    stamp(3);
    tix++;
    work = next;
#pragma unroll
    for (int p = 0; p < NPA; p++) vAc[p] = vAn[p];
#pragma unroll
    for (int p = 0; p < NPB; p++) vBc[p] = vBn[p];
    m0c = m0n; n0c = n0n; b0c = b0n;
    if constexpr (SK) uc = un;
    par ^= 1;
  }
#endif
}

template <int TM, int TP, int WM, int WP, bool F32, bool BATCHED = false>
int launch(Args a, hipStream_t st, int batch = 1) {
  a.ntn = (a.N + TM - 1) / TM;
  a.tpb = ((a.M + TP - 1) / TP) * a.ntn;
  a.total = a.tpb * batch;
  DMVAE_CHECK_ARG(a.total > 0 && a.total < (1 << 24) && (long long)a.tpb * batch < (1 << 24),
                  "linear_bf16: %lld output tiles (the kernel's tile index arithmetic is exact below 2^24)", (long long)a.tpb * batch);
  a.inv_ntn = 1.0f / (float)a.ntn;
  a.inv_tpb = 1.0f / (float)a.tpb;
  const unsigned grid = a.total > 256 ? 256u : (unsigned)a.total;
  constexpr int slot = (TM + TP) * 64;
  constexpr int fit = (160 * 1024 - 3 * 1024) / slot;          // ring slots that fit beside the two bias slots and the dump KiB
  constexpr int nbuf = fit > GEMM_MAXBUF ? GEMM_MAXBUF : fit;
  static_assert(nbuf >= 4, "LDS");
  constexpr int lds = nbuf * slot + 3 * 1024;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_pp_kernel<TM, TP, WM, WP, F32, nbuf, BATCHED>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done = true;
  }
  hipLaunchKernelGGL((gemm_pp_kernel<TM, TP, WM, WP, F32, nbuf, BATCHED>), dim3(grid), dim3(WM * WP * 64), lds, st, a);
  DMVAE_CHECK_LAUNCH();
  return 0;
}
// SK launch.  Tile 0: 256 x 256 (ring 4 deep: parts of >= 8 K steps; 256 KB of f32 per partial tile).  Tile 1: 256 columns x 128 rows (ring 6 deep: parts of >= 12
// steps; 128 KB per partial) -- the vendor library's stream-K solutions for these shapes run 128 x 160 macro-tiles (profiles/r6_hipblaslt_kernel_choices_probe.txt):
// a cut costs a third of the partial bytes of a 256 x 256 tile, and 144 tiles x 96 steps spread over 256 CUs are 54 steps each instead of 96.
constexpr int SK_GRID = 256, SK_MAX_TILES = 2048, SK_COUNTER_BYTES = SK_MAX_TILES * 8 * 4;
struct SkTile { int tm, tp, minseg; };
static inline SkTile sk_tile(int tile) { return tile == 1 ? SkTile{256, 128, 12} : SkTile{256, 256, 8}; }   // minseg = 2 * (ring depth - 1) + 2
template <int TM, int TP, int WM, int WP>
static int launch_sk_t(Args a, int splits, hipStream_t st) {
  a.ntn = (a.N + TM - 1) / TM;
  a.tpb = ((a.M + TP - 1) / TP) * a.ntn;
  a.total = a.tpb;
  a.inv_ntn = 1.0f / (float)a.ntn;
  a.inv_tpb = 1.0f / (float)a.tpb;
  a.skS = splits;
  a.V = splits > 0 ? a.total * splits : SK_GRID;
  a.T = a.total * (a.K >> 5);
  const unsigned grid = a.V > SK_GRID ? (unsigned)SK_GRID : (unsigned)a.V;
  constexpr int slot = (TM + TP) * 64;
  constexpr int fit = (160 * 1024 - 3 * 1024) / slot;
  constexpr int nbuf = fit > GEMM_MAXBUF ? GEMM_MAXBUF : fit;
  constexpr int lds = nbuf * slot + 3 * 1024;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_pp_kernel<TM, TP, WM, WP, false, nbuf, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done = true;
  }
  hipLaunchKernelGGL((gemm_pp_kernel<TM, TP, WM, WP, false, nbuf, false, true>), dim3(grid), dim3(512), lds, st, a);
  DMVAE_CHECK_LAUNCH();
  return 0;
}
static int launch_sk(Args a, int splits, int tile, hipStream_t st) {
  return tile == 1 ? launch_sk_t<256, 128, 4, 2>(a, splits, st) : launch_sk_t<256, 256, 2, 4>(a, splits, st);
}
// What dmvae_linear_bf16_sk needs of (M, N, K, splits, tile): splits 0 = stream-K (ranges of T / 256 steps, at least 2 MINSEG each so that no cut collapses), else
// uniform parts of >= MINSEG steps.
static bool sk_ok(int M, int N, int K, int splits, int tile) {
  if (!(M >= 64 && N > 0 && N % 8 == 0 && K % 32 == 0 && K >= 384 && splits >= 0 && splits <= 8 && (tile == 0 || tile == 1))) return false;
  const SkTile t = sk_tile(tile);
  const long long tiles = (long long)((M + t.tp - 1) / t.tp) * ((N + t.tm - 1) / t.tm);
  const int nK = K >> 5;
  if (tiles > SK_MAX_TILES || tiles * nK >= (1ll << 30)) return false;
  if (splits == 0) return tiles * nK / SK_GRID >= 2 * t.minseg && nK >= 2 * t.minseg;
  return nK / splits >= t.minseg && tiles * splits <= 4096;
}

// The tile menu: (columns, rows, cost of one tile relative to a 256 x 256 tile's at the same K with the whole chip busy -- measured on the 16384 x 6144 x 1152
// problem, 6 to 12 rounds per entry: tools/bench_gemm.py --sweep --cold --kmajor).  Smaller tiles cost more per flop: 0.80 for 62.5 % of the area.
struct Cfg { int tm, tp; float cost; };
static const Cfg g_cfg[] = {
    {256, 256, 1.00f}, {256, 224, 0.94f}, {256, 192, 0.885f}, {256, 160, 0.80f}, {256, 128, 0.67f},
    {128, 448, 1.02f}, {128, 384, 0.93f}, {128, 256, 0.70f},  {192, 256, 0.90f}, {192, 320, 1.12f},
    // round 5: the M = 4096 .. 4112 problems of the DMD stage (batch 16: LightningDiT-XL/1's N = 1152 Linears and their input gradients, ViT-L's N = 1024 ones) are
    // 160 / 132 tiles of 256 x 128 -- half to two thirds of the chip; these split them into 192 / 198 smaller tiles
    {192, 128, 0.54f}, {128, 192, 0.54f},
};
constexpr int NCFG = sizeof(g_cfg) / sizeof(g_cfg[0]);

static int g_forced = -1;   // -1: plan by cost; >= 0: this menu entry (dmvae_debug_gemm_cfg: tools/bench_gemm.py's sweep, tests/test_gpu_gemm_pp.py)
// Time model: tiles / 256 rounds of the tile's cost -- FRACTIONAL rounds, because the chip is power-limited: with half of the CUs idle in the last round the
// busy ones clock higher and the round ends sooner (896 tiles of 256 x 256 = 3.5 rounds measure 3.5 x one round's time, not 4 x) -- but a round never costs
// less than 0.85 of a full one.  Ragged tiles are whole tiles (their padding rows / columns cost what real ones do).
static int plan(int M, int N, int K, bool gated = false) {
  const int forced = g_forced;
  if (forced >= 0 && forced < NCFG && !(gated && g_cfg[forced].tm == 192)) return forced;
  (void)K;
  // the two round-5 tiles (entries 10, 11) are in the menu but not in the plan by default: isolated (tools/bench_gemm.py --sweep --cold, profiles/r5_gemm_sweep_m4096.txt) they
  // win ViT-L's N = 1024 Linears at 16 x 257 tokens by 13-17 %, inside the DMD stage's step the same launches ran 1.2 ms per step SLOWER than on 256 x 128
  // (profiles/r5_dmd_tiles_ab.txt) -- DMVAE_GEMM_NPLAN=12 plans over them
  static const int nplan = [] { const char* e = getenv("DMVAE_GEMM_NPLAN"); const int v = e ? atoi(e) : 10; return v < 1 ? 1 : (v > NCFG ? NCFG : v); }();
  int best = 0;
  float best_t = 1e30f;
  for (int c = 0; c < nplan; c++) {
    if (gated && g_cfg[c].tm == 192) continue;   // the gated epilogue pairs the halves of a lane's 4 or 8 columns
    const long long tiles = (long long)((M + g_cfg[c].tp - 1) / g_cfg[c].tp) * ((N + g_cfg[c].tm - 1) / g_cfg[c].tm);
    const float frac = (float)tiles / 256.0f, whole = 0.85f * (float)((tiles + 255) / 256);
    const float t = (frac > whole ? frac : whole) * g_cfg[c].cost;
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
    case 5: return launch<128, 448, 2, 4, F32>(a, st);
    case 6: return launch<128, 384, 2, 4, F32>(a, st);
    case 7: return launch<128, 256, 2, 4, F32>(a, st);
    case 8: return launch<192, 256, 2, 4, F32>(a, st);
    case 9: return launch<192, 320, 2, 4, F32>(a, st);
    case 10: return launch<192, 128, 2, 4, F32>(a, st);
    default: return launch<128, 192, 2, 4, F32>(a, st);
  }
}

}  // namespace dmvae_gemm_pp

static unsigned long long* g_gemm_dbg = nullptr;
extern "C" void dmvae_debug_gemm_timing(void* buf) { g_gemm_dbg = (unsigned long long*)buf; }   // diagnostics only (tools/probes/time_gemm_pp.py)
extern "C" void dmvae_debug_gemm_cfg(int cfg) { dmvae_gemm_pp::g_forced = cfg; }   // diagnostics only: force a menu entry (0 .. 9), -1 = plan by cost

extern "C" int dmvae_linear_bf16_plan(int M, int N, int K, int* tile_cols, int* tile_rows) {
  using namespace dmvae_gemm_pp;
  const int c = plan(M, N, K);
  if (tile_cols) *tile_cols = g_cfg[c].tm;
  if (tile_rows) *tile_rows = g_cfg[c].tp;
  return c;
}

// `batch` independent products y_b [M][ldy] = x_b [M][lda] w_b [N][ldw]^T on the same kernel (BATCHED instantiations): the decoder attention's per-sample GEMMs
// (flux_ae.py:37-49: q k^T, p v, and their input gradients), 16 to 8 tiles of 256 x 256 per sample.  Element strides between the products; no bias, no activation.
extern "C" int dmvae_linear_bf16_batched_supported(int batch, int M, int N, int K) {
  return (batch > 0 && M >= 64 && N > 0 && N % 8 == 0 && K >= 384 && K % 32 == 0) ? 1 : 0;
}
extern "C" int dmvae_linear_bf16_batched(const void* x, const void* w, void* y, int batch, int M, int N, int K, int lda, int ldw, int ldy, long long sx, long long sw,
                                         long long sy, int out_f32, hipStream_t stream) {
  using namespace dmvae_gemm_pp;
  DMVAE_CHECK_ARG(x && w && y, "linear_bf16_batched: null operand");
  DMVAE_CHECK_ARG(dmvae_linear_bf16_batched_supported(batch, M, N, K), "linear_bf16_batched: need K %% 32 == 0, K >= 384, N %% 8 == 0, M >= 64 (batch %d, M %d, N %d, K %d)",
                  batch, M, N, K);
  DMVAE_CHECK_ARG(lda >= K && ldw >= K && ldy >= N && lda % 8 == 0 && ldw % 8 == 0 && ldy % 8 == 0, "linear_bf16_batched: leading dimensions must cover the rows and be multiples of 8");
  DMVAE_CHECK_ARG(sx >= (long long)M * lda && sw >= (long long)N * ldw && sy >= (long long)M * ldy && sx % 8 == 0 && sw % 8 == 0 && sy % 8 == 0,
                  "linear_bf16_batched: batch strides must cover one product's operand and be multiples of 8 elements");
  const long long xb = ((long long)(batch - 1) * sx + (long long)M * lda) * 2, wb = ((long long)(batch - 1) * sw + (long long)N * ldw) * 2;
  const long long yb = ((long long)(batch - 1) * sy + (long long)M * ldy) * (out_f32 ? 4 : 2);
  DMVAE_CHECK_ARG(xb < (1ll << 31) && wb < (1ll << 31) && yb < (1ll << 31), "linear_bf16_batched: operands are addressed through 32-bit buffer offsets (2 GiB each over the whole batch)");
  Args a;
  a.x = (const bf16*)x; a.w = (const bf16*)w; a.bias = nullptr; a.y = y;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldw = ldw; a.ldy = ldy;
  a.act = 0; a.bias_bf16 = 0; a.H = N / 2; a.ntn = 0; a.total = 0; a.inv_ntn = 0.f; a.dbg = nullptr;
  a.V = a.skS = a.T = 0; a.slabs = nullptr; a.counters = nullptr; a.slab_bytes = 0u; a.y2 = nullptr; a.ldy2 = 0;
  a.wbytes = (unsigned)wb; a.xbytes = (unsigned)xb; a.ybytes = (unsigned)yb;
  a.wsRow = (unsigned)ldw * 2u; a.wsK = 64u;
  a.sA = (unsigned)(sw * 2); a.sB = (unsigned)(sx * 2); a.sY = (unsigned)(sy * (out_f32 ? 4 : 2));
  // three of the menu's tiles: rounds x cost over the whole batch's tiles (plan()'s model)
  static const int cand[3] = {0, 4, 7};
  int best = 0;
  float best_t = 1e30f;
  for (int i = 0; i < 3; i++) {
    const Cfg& c = g_cfg[cand[i]];
    const long long tiles = (long long)batch * ((M + c.tp - 1) / c.tp) * ((N + c.tm - 1) / c.tm);
    const float frac = (float)tiles / 256.0f, whole = 0.85f * (float)((tiles + 255) / 256);
    const float t = (frac > whole ? frac : whole) * c.cost;
    if (t < best_t * 0.999f) { best_t = t; best = cand[i]; }
  }
  if (out_f32) {
    if (best == 0) return launch<256, 256, 2, 4, true, true>(a, stream, batch);
    if (best == 4) return launch<256, 128, 4, 2, true, true>(a, stream, batch);
    return launch<128, 256, 2, 4, true, true>(a, stream, batch);
  }
  if (best == 0) return launch<256, 256, 2, 4, false, true>(a, stream, batch);
  if (best == 4) return launch<256, 128, 4, 2, false, true>(a, stream, batch);
  return launch<128, 256, 2, 4, false, true>(a, stream, batch);
}

// Split-K form for few-tile, deep-K problems (LightningDiT at batch 16: M = 4096, N = 1152 is 18 tiles of 256 x 256 -- w3's K = 3072, the input gradients of qkv /
// w12 with K = 3456 / 6144): the reduction is cut into `splits` equal parts, each part is one "product" of the BATCHED 256 x 256 instantiation (its operand offsets
// select the K range: x by splits * 64-byte columns, w by the same columns (row-major) or by whole K tiles (K-tile-major); its y offset the part's f32 slab), so that
// 80 tiles become 240 work units of a third of the length.  slabs: f32 [splits][M][N]; the caller sums them (dmvae_splitk_sum_bf16) -- fixed order, deterministic.
extern "C" int dmvae_linear_bf16_splitk_supported(int M, int N, int K, int splits) {
  return (splits >= 2 && splits <= 8 && M >= 64 && N > 0 && N % 8 == 0 && K % (32 * splits) == 0 && K / splits >= 384 &&
          (long long)splits * M * N * 4 < (1ll << 31)) ? 1 : 0;
}
extern "C" int dmvae_linear_bf16_splitk(const void* x, const void* w, void* slabs, int splits, int M, int N, int K, int lda, int ldw, int w_layout, hipStream_t stream) {
  using namespace dmvae_gemm_pp;
  DMVAE_CHECK_ARG(x && w && slabs, "linear_bf16_splitk: null operand");
  DMVAE_CHECK_ARG(dmvae_linear_bf16_splitk_supported(M, N, K, splits), "linear_bf16_splitk: M %d N %d K %d splits %d (K %% (32 splits) == 0, K / splits >= 384, N %% 8 == 0)", M, N, K, splits);
  DMVAE_CHECK_ARG(w_layout == 0 || w_layout == 1, "linear_bf16_splitk: w_layout must be 0 or 1");
  DMVAE_CHECK_ARG(lda >= K && (w_layout == 1 || ldw >= K) && lda % 8 == 0 && ldw % 8 == 0, "linear_bf16_splitk: leading dimensions must cover the rows and be multiples of 8");
  const long long wb = w_layout == 1 ? (long long)N * K * 2 : (long long)N * ldw * 2;
  DMVAE_CHECK_ARG((long long)M * lda * 2 < (1ll << 31) && wb < (1ll << 31), "linear_bf16_splitk: operands are addressed through 32-bit buffer offsets (2 GiB each)");
  const int Ks = K / splits;
  Args a;
  a.x = (const bf16*)x; a.w = (const bf16*)w; a.bias = nullptr; a.y = slabs;
  a.M = M; a.N = N; a.K = Ks; a.lda = lda; a.ldw = ldw; a.ldy = N;
  a.act = 0; a.bias_bf16 = 0; a.H = N / 2; a.ntn = 0; a.total = 0; a.inv_ntn = 0.f; a.dbg = nullptr;
  a.V = a.skS = a.T = 0; a.slabs = nullptr; a.counters = nullptr; a.slab_bytes = 0u; a.y2 = nullptr; a.ldy2 = 0;
  a.wbytes = (unsigned)wb; a.xbytes = (unsigned)((long long)M * lda * 2); a.ybytes = (unsigned)((long long)splits * M * N * 4);
  a.wsRow = w_layout == 1 ? 64u : (unsigned)ldw * 2u;
  a.wsK = w_layout == 1 ? (unsigned)N * 64u : 64u;
  a.sB = (unsigned)Ks * 2u;                                                      // x: Ks columns further per part
  a.sA = w_layout == 1 ? (unsigned)(Ks / 32) * (unsigned)N * 64u : (unsigned)Ks * 2u;   // w: whole K tiles (K-tile-major) or Ks columns (row-major)
  a.sY = (unsigned)((long long)M * N * 4);
  return launch<256, 256, 2, 4, true, true>(a, stream, splits);
}
// y bf16 [n] = bf16(sum_s slabs[s][n] + bias[col]) (bias f32 / bf16 [N] or null; n = M * N): the fixed-order sum of dmvae_linear_bf16_splitk's parts, rounded once.
namespace dmvae_gemm_pp {
__global__ __launch_bounds__(256) void splitk_sum_kernel(const float* __restrict__ slabs, const void* __restrict__ bias, int bias_bf16, bf16* __restrict__ y, size_t n8,
                                                         size_t slab, int S, int N) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    f32x4 a0 = *reinterpret_cast<const f32x4*>(slabs + i * 8), a1 = *reinterpret_cast<const f32x4*>(slabs + i * 8 + 4);
    for (int s = 1; s < S; s++) {
      a0 += *reinterpret_cast<const f32x4*>(slabs + s * slab + i * 8);
      a1 += *reinterpret_cast<const f32x4*>(slabs + s * slab + i * 8 + 4);
    }
    if (bias) {
      const int c = (int)((i * 8) % (size_t)N);
#pragma unroll
      for (int e = 0; e < 4; e++) {
        a0[e] += bias_bf16 ? (float)((const bf16*)bias)[c + e] : ((const float*)bias)[c + e];
        a1[e] += bias_bf16 ? (float)((const bf16*)bias)[c + 4 + e] : ((const float*)bias)[c + 4 + e];
      }
    }
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 4; e++) { o[e] = (bf16)a0[e]; o[4 + e] = (bf16)a1[e]; }
    *reinterpret_cast<bf16x8*>(y + i * 8) = o;
  }
}
}  // namespace dmvae_gemm_pp
extern "C" int dmvae_splitk_sum_bf16(const void* slabs, int splits, const void* bias, int bias_bf16, void* y, int M, int N, hipStream_t stream) {
  using namespace dmvae_gemm_pp;
  DMVAE_CHECK_ARG(slabs && y && splits >= 1 && M > 0 && N > 0 && N % 8 == 0, "splitk_sum_bf16: bad argument (N %% 8 == 0)");
  const size_t n8 = (size_t)M * N / 8;
  const unsigned grid = (unsigned)((n8 + 255) / 256 > 4096 ? 4096 : (n8 + 255) / 256);
  hipLaunchKernelGGL(splitk_sum_kernel, dim3(grid), dim3(256), 0, stream, (const float*)slabs, bias, bias_bf16, (bf16*)y, n8, (size_t)M * N, splits, N);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

// Stream-K / fused split-K Linear (the SK instantiation above): y bf16 = act(x w^T + bias) with the reduction cut across workgroups and summed in K order by the
// last part to arrive -- one launch, no slab pass.  splits = 0: stream-K (256 equal ranges of the (tile, K step) space: the cut depends on M); splits >= 2: that
// many uniform parts per tile (depends on N, K only: a row's bits do not depend on how many rows the call has).  workspace: dmvae_linear_bf16_sk_workspace bytes,
// whose FIRST dmvae_linear_bf16_sk_counter_bytes() bytes must be zero on entry (the kernel leaves them zero; zero the buffer once).
extern "C" int dmvae_linear_bf16_sk_supported(int M, int N, int K, int splits, int tile) { return dmvae_gemm_pp::sk_ok(M, N, K, splits, tile) ? 1 : 0; }
extern "C" size_t dmvae_linear_bf16_sk_counter_bytes(void) { return (size_t)dmvae_gemm_pp::SK_COUNTER_BYTES; }
extern "C" size_t dmvae_linear_bf16_sk_workspace(int M, int N, int K, int splits, int tile) {
  using namespace dmvae_gemm_pp;
  if (!sk_ok(M, N, K, splits, tile)) return 0;
  const SkTile t = sk_tile(tile);
  const long long tiles = (long long)((M + t.tp - 1) / t.tp) * ((N + t.tm - 1) / t.tm);
  const long long V = splits > 0 ? tiles * splits : SK_GRID;
  return (size_t)SK_COUNTER_BYTES + (size_t)(2 * V) * t.tm * t.tp * 4;
}
extern "C" int dmvae_linear_bf16_sk(const void* x, const void* w, const void* bias, void* y, void* workspace, size_t workspace_bytes, int splits, int tile, int M, int N,
                                    int K, int lda, int ldw, int ldy, int act, int bias_bf16, int w_layout, hipStream_t stream) {
  using namespace dmvae_gemm_pp;
  DMVAE_CHECK_ARG(x && w && y && workspace, "linear_bf16_sk: null operand");
  DMVAE_CHECK_ARG(sk_ok(M, N, K, splits, tile), "linear_bf16_sk: M %d N %d K %d splits %d tile %d not taken (dmvae_linear_bf16_sk_supported)", M, N, K, splits, tile);
  DMVAE_CHECK_ARG(w_layout == 0 || w_layout == 1, "linear_bf16_sk: w_layout must be 0 (row-major [N][ldw]) or 1 (K-tile-major [K / 32][N][32])");
  DMVAE_CHECK_ARG(lda >= K && (w_layout == 1 || ldw >= K) && ldy >= (act == 6 ? N / 2 : N) && lda % 8 == 0 && ldw % 8 == 0 && ldy % 8 == 0,
                  "linear_bf16_sk: leading dimensions must cover the rows and be multiples of 8");
  DMVAE_CHECK_ARG(act == 0 || act == 1 || act == 5 || act == 6, "linear_bf16_sk: act must be 0 (none), 1 (SiLU), 5 (GELU) or 6 (SwiGLU over the [x1 | x2] halves of N)");
  DMVAE_CHECK_ARG(act != 6 || N % 16 == 0, "linear_bf16_sk: the SwiGLU epilogue needs N %% 16 == 0");
  const long long wb = w_layout == 1 ? (long long)N * K * 2 : (long long)N * ldw * 2;
  DMVAE_CHECK_ARG((long long)M * lda * 2 < (1ll << 31) && wb < (1ll << 31) && (long long)M * ldy * 2 < (1ll << 31),
                  "linear_bf16_sk: operands are addressed through 32-bit buffer offsets (2 GiB each)");
  const size_t need = dmvae_linear_bf16_sk_workspace(M, N, K, splits, tile);
  DMVAE_CHECK_ARG(workspace_bytes >= need && need - SK_COUNTER_BYTES < (1ull << 32), "linear_bf16_sk: workspace too small (need %zu bytes, see dmvae_linear_bf16_sk_workspace)", need);
  Args a;
  a.x = (const bf16*)x; a.w = (const bf16*)w; a.bias = bias; a.y = y;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldw = ldw; a.ldy = ldy;
  a.act = act; a.bias_bf16 = bias_bf16; a.H = N / 2; a.ntn = 0; a.total = 0; a.inv_ntn = 0.f; a.dbg = g_gemm_dbg;
  a.wbytes = (unsigned)wb; a.tpb = 0; a.inv_tpb = 0.f; a.sA = a.sB = a.sY = a.xbytes = a.ybytes = 0u;
  a.wsRow = w_layout == 1 ? 64u : (unsigned)ldw * 2u;
  a.wsK = w_layout == 1 ? (unsigned)N * 64u : 64u;
  a.y2 = nullptr; a.ldy2 = 0;
  a.counters = (unsigned*)workspace;
  a.slabs = (float*)((char*)workspace + SK_COUNTER_BYTES);
  a.slab_bytes = (unsigned)(need - SK_COUNTER_BYTES);
  return launch_sk(a, splits, tile, stream);
}

static int linear_bf16_impl(const void* x, const void* w, const void* bias, void* y, void* y2, int ldy2, int M, int N, int K, int lda, int ldw, int ldy,
                            int act, int bias_bf16, int out_f32, int w_layout, hipStream_t stream);
extern "C" int dmvae_linear_bf16(const void* x, const void* w, const void* bias, void* y, int M, int N, int K, int lda, int ldw, int ldy,
                                 int act, int bias_bf16, int out_f32, int w_layout, hipStream_t stream) {
  return linear_bf16_impl(x, w, bias, y, nullptr, 0, M, N, K, lda, ldw, ldy, act, bias_bf16, out_f32, w_layout, stream);
}
// SwiGLU FFN's first half in one launch: x12 [M][ldx12] = bf16(x w^T + bias) over the N = 2 H columns [x1 | x2] AND g [M][ldg] = silu(x1) * x2 over H columns --
// what dmvae_linear_bf16 (act 0) followed by dmvae_swiglu_bf16 writes, bit for bit (the product is formed from the bf16-rounded halves), without the second pass
// over the 2 H-wide tensor.  x12 is what the backward of SwiGLU reads.  Reference: swiglu_ffn.py:31-36 (w12, chunk, silu(x1) * x2).
extern "C" int dmvae_linear_bf16_swiglu_pre(const void* x, const void* w, const void* bias, void* g, void* x12, int M, int N, int K, int lda, int ldw, int ldg,
                                            int ldx12, int bias_bf16, int w_layout, hipStream_t stream) {
  DMVAE_CHECK_ARG(x12 && ldx12 >= N && ldx12 % 8 == 0, "linear_bf16_swiglu_pre: x12 must be given with ldx12 >= N, a multiple of 8");
  return linear_bf16_impl(x, w, bias, g, x12, ldx12, M, N, K, lda, ldw, ldg, 6, bias_bf16, 0, w_layout, stream);
}
static int linear_bf16_impl(const void* x, const void* w, const void* bias, void* y, void* y2, int ldy2, int M, int N, int K, int lda, int ldw, int ldy,
                            int act, int bias_bf16, int out_f32, int w_layout, hipStream_t stream) {
  using namespace dmvae_gemm_pp;
  DMVAE_CHECK_ARG(x && w && y, "linear_bf16: null operand");
  DMVAE_CHECK_ARG(M > 0 && N > 0 && K >= 384 && K % 32 == 0 && N % 8 == 0, "linear_bf16: need K %% 32 == 0, K >= 384 and N %% 8 == 0 (M %d, N %d, K %d)", M, N, K);
  DMVAE_CHECK_ARG(w_layout == 0 || w_layout == 1, "linear_bf16: w_layout must be 0 (row-major [N][ldw]) or 1 (K-tile-major [K / 32][N][32])");
  DMVAE_CHECK_ARG(lda >= K && (w_layout == 1 || ldw >= K) && ldy >= (act == 6 ? N / 2 : N) && lda % 8 == 0 && ldw % 8 == 0 && ldy % 8 == 0,
                  "linear_bf16: leading dimensions must cover the rows and be multiples of 8");
  DMVAE_CHECK_ARG(act == 0 || act == 1 || act == 5 || act == 6, "linear_bf16: act must be 0 (none), 1 (SiLU), 5 (GELU) or 6 (SwiGLU over the [x1 | x2] halves of N)");
  DMVAE_CHECK_ARG(act != 6 || (!out_f32 && N % 16 == 0), "linear_bf16: the SwiGLU epilogue writes bf16 and needs N %% 16 == 0");
  const long long wb = w_layout == 1 ? (long long)N * K * 2 : (long long)N * ldw * 2;
  DMVAE_CHECK_ARG((long long)M * lda * 2 < (1ll << 31) && wb < (1ll << 31) && (long long)M * ldy * (out_f32 ? 4 : 2) < (1ll << 31),
                  "linear_bf16: operands are addressed through 32-bit buffer offsets (2 GiB each)");
  Args a;
  a.x = (const bf16*)x; a.w = (const bf16*)w; a.bias = bias; a.y = y;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldw = ldw; a.ldy = ldy;
  a.act = act; a.bias_bf16 = bias_bf16; a.H = N / 2; a.ntn = 0; a.total = 0; a.inv_ntn = 0.f; a.dbg = g_gemm_dbg;
  a.V = a.skS = a.T = 0; a.slabs = nullptr; a.counters = nullptr; a.slab_bytes = 0u; a.y2 = y2; a.ldy2 = ldy2;
  DMVAE_CHECK_ARG(!y2 || (long long)M * ldy2 * 2 < (1ll << 31), "linear_bf16_swiglu_pre: x12 is addressed through 32-bit buffer offsets (2 GiB)");
  a.wbytes = (unsigned)wb; a.tpb = 0; a.inv_tpb = 0.f; a.sA = a.sB = a.sY = a.xbytes = a.ybytes = 0u;
  a.wsRow = w_layout == 1 ? 64u : (unsigned)ldw * 2u;
  a.wsK = w_layout == 1 ? (unsigned)N * 64u : 64u;
  const int cfg = plan(M, N, K, act == 6);
  return out_f32 ? dispatch<true>(cfg, a, stream) : dispatch<false>(cfg, a, stream);
}
