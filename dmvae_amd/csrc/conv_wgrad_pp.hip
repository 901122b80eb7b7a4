// Weight gradient of the NHWC bf16 convolution on gfx950 -- "ping-pong" kernel for the large decoder layers.
//
// Same contract as conv_wgrad.hip (autograd's conv / linear weight gradient for nn.Conv2d at models/flux_ae.py:32-35,
// 63,65,67,101 and nn.Linear at models/vae.py:58-62):
//
//   dW[co][tap][ci] = sum_p dy[p][co] * a[p (+) tap][ci]                  (reduction over pixels, split-K, deterministic)
//
// GEMM view: M = cout, N = (tap, cin) flattened in column groups of 128 channels, K = pixels.  Both operands are
// channel-contiguous while the reduction runs over pixels, so LDS holds [32 pixels][128 channels] sub-tiles exactly as
// DMA'd and the MFMA fragments come from the gfx950 transpose read ds_read_b64_tr_b16 (hardware-checked by
// tools/probes/probe_tr16.hip).  The MFMA is v_mfma_f32_16x16x32_bf16 (one instruction = 16 x 16 outputs over the whole 32-pixel K tile;
// same flops for ~5 % less power than 32x32x16 in the power-limited regime, see conv_pp.hip): lane group G = lane >> 4 supplies pixels
// 8 G .. 8 G + 7, so the two groups a transpose read serves per LDS pass ({0,1} or {2,3}) touch pixel rows {0-3, 8-11} (+4, +16) of the same
// 16-channel block.  The LDS image is therefore XOR-swizzled in 32-B slots keyed on ((pixel & 3) << 1) | ((pixel >> 3) & 1) -- eight
// distinct slots for those eight rows, 256 B per pass -- applied on the DMA source address and on the read.
//
// Structure = conv_pp.hip: 8 waves, each 128x64 (or 64x96) of the 256x256 (or 128x384) output tile, a 4-deep LDS ring
// filled by LDS-DMA through buffer descriptors three K tiles ahead (counted vmcnt), the two waves of a SIMD alternating
// LOAD / COMPUTE intervals between two s_barrier per K tile.  The 128x384 configuration serves Cout or Cin = 128: its
// three column groups are three different taps (or tap x cin-half), so a 128-channel layer still stages ~96 FLOP/B.
//
// A K tile is 32 consecutive pixels of one image row (the host only selects this kernel when W % 32 == 0), so the
// zero-padding test is wave-uniform except for the first / last pixel of a row: per-lane sources are fixed for the
// whole kernel and only the wave-uniform soffset advances.
//
// S2 instantiation: the 4x4 stride-2 padding-1 conv (models/patchgan.py:125-133; and, with the operands' roles exchanged, the weight gradient of Upsample's
// conv in its sub-pixel form -- include/dmvae_hip.h, dmvae_subpixel_weight): tap (ky, kx) of output pixel (y, x) reads source (2y - 1 + ky, 2x - 1 + kx), so
// a K tile's 32 pixels sit two source pixels apart (per-lane offsets doubled) and the row test reads 2y - 1 + ky; sixteen taps instead of nine.
#include <algorithm>
#include <vector>
#include "common.h"
#include "dmvae_hip.h"
#include <cstdlib>
constexpr int WG_NBUF = 4;   // ring depth (K tiles in LDS), both forms

namespace dmvae_wgrad_pp {

struct Args {
  const bf16* dy;  // [M, Cout]
  const bf16* a;   // [N, Hi, Wi, Cin]
  float* slab;     // [splits][Cout][T][Cin]
  float* bslab;    // [splits][Cout] bias-gradient partials (column sums of dy) or null
  int N, Hi, Wi, Cin, Ho, Wo, Cout;
  int ks, ups, M, kchunk;  // kchunk: pixels per split (multiple of 32)
  int mtiles, ntiles, ngroups, gpt;  // gpt: 128-channel column groups per cin row (Cin / 128)
};

constexpr unsigned SENT = 0x80000000u;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  // the builtin, not inline asm, so that the compiler's wait-count pass sees it (see conv_pp.hip: otherwise it drains the queue with vmcnt(0) every K tile)
  __builtin_amdgcn_s_waitcnt((N & 15) | 0x0F70 | ((N >> 4) << 14));
  asm volatile("" ::: "memory");
}
// Wave-uniform values that come out of an integer division sit in VGPRs (the division runs on the VALU); everything derived from them then
// stays there and every LDS-DMA issue whose soffset depends on them becomes a readfirstlane waterfall loop -- four per K tile, plus a
// compiler-inserted vmcnt(0) at the loop head because its wait-count model cannot count loads inside those loops.  Pin them to SGPRs.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// Transpose read as inline asm, not __builtin_amdgcn_ds_read_tr16_b64: the compiler's wait-count pass orders every LDS read it can see behind every earlier LDS-DMA
// (it cannot prove the ring slots disjoint) and put an s_waitcnt vmcnt(0) at the head of the K loop -- the three-tile prefetch queue was drained once per K tile and
// the loop ran at the latency of the newest piece (tools/loop_waits.py shows the skeleton; 128->128 @256^2: 815 -> see DESIGN_HISTORY.md 8.12).  The asm form carries no memory
// operand; the loop's own counted vmcnt + barrier protocol is what orders the reads behind the pieces they need, and lgkmcnt(0) ahead of the barrier covers the results.
template <int OFF>
__device__ __forceinline__ s16x4 tr_read(unsigned lds_addr) {
  s16x4 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(lds_addr), "n"(OFF));
  return r;
}

// GA / GB: 128-channel sub-tiles of the dy / activation operand per block; waves WM x WN.
// HALO (3x3 stride 1, GB = 3): the block's three column groups are the taps kx = 0, 1, 2 of ONE (ky, 128-channel slice), and the activation operand of a K tile is
// staged once as the 34 pixels x0 - 1 .. x0 + 32 of source row y + ky - 1 (36 LDS rows, 9 pieces) instead of three shifted copies of 32 (24 pieces): a tap's
// fragment is the same transpose read one pixel row further on.  Per K tile the block then stages 17 KiB instead of 32 -- the 128 x 384 tile runs 24 MFMAs per
// wave between barriers against the 256 x 256 tile's 32 with the same four DMA issues per wave, and its LOAD interval, not its COMPUTE interval, set the pace.
// KP = 64 (HALO only): a K tile is 64 pixels -- two 32-pixel halves of the dy operand and one 66-pixel halo tile (68 LDS rows, 17 pieces); 48 MFMAs per wave between
// barriers instead of 24 against 40 fragment reads and 4-5 DMA issues: the 24-MFMA COMPUTE interval (384 cycles) sat under a ~600-cycle LOAD interval.
// RAGGED (the grouped Linear form only): M need not be a multiple of 32 -- the rows of the last K tile past M are masked per lane (the descriptors' range check does
// not see the scalar K-tile offset, so it cannot do it)
template <int GA, int GB, int WM, int WN, bool S2, bool HALO, bool UPS, int KP, bool RAGGED>
__device__ __forceinline__ void wgrad_pp_body(const Args& a, const unsigned flat_block, const unsigned total_blocks) {
#if __HIP_DEVICE_COMPILE__
  constexpr int TM = GA * 128, TN = GB * 128;
  constexpr int BM = TM / WM / 16, BN = TN / WN / 16;  // 16 x 16 output blocks per wave
  constexpr int SUB = 32 * 256;  // bytes of one [32 px][128 ch] sub-tile
  constexpr int KH = KP / 32;    // 32-pixel halves per K tile
  constexpr int HROWS = KP + 4;  // HALO: LDS rows of the halo tile (KP + 2 pixels, padded to whole 4-row pieces)
  constexpr int BSZ = HALO ? HROWS * 256 : GB * SUB;
  constexpr int ASZ = GA * KH * SUB;
  constexpr int SLOT = ASZ + BSZ;
  constexpr int NBUF = HALO ? (KH == 2 ? 4 : WG_NBUF) : WG_NBUF, PF = NBUF - 1;
  constexpr int NQ = HROWS / 4, FP = (NQ - 1) / 8;   // HALO: pieces of the halo tile; FP per wave + the last one, which is wave 7's
  constexpr int NPA = GA * KH, NPB = HALO ? FP + 1 : GB;  // 1-KiB pieces per wave per K tile (8 pieces per sub-tile, 8 waves)
  constexpr int NP = NPA + NPB;
  static_assert(WM * WN == 8, "8 waves");
  static_assert(!HALO || (GB == 3 && !S2), "HALO: three taps of one kernel row");
  static_assert(KP == 32 || (KP == 64 && HALO && GA == 1), "64-pixel K tiles: the halo form only");
  static_assert(!UPS || (!S2 && !HALO), "UPS: nearest-x2 source walk of the plain 3x3 form");   // compile-time: the run-time test put two exec-masked blocks into every K tile
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const int wm = wave / WN, wn = wave % WN;
  const int T = a.ks * a.ks;

  const unsigned wid = (unsigned)uni((int)(total_blocks ? xcd_remap(flat_block, total_blocks) : flat_block));   // total_blocks == 0: the caller placed this block itself (grouped launch by XCD)
  const int tiles = a.mtiles * a.ntiles;
  const int split = uni((int)(wid / tiles));
  const int tile = uni((int)(wid - (unsigned)split * tiles));
  const int mt = uni(tile / a.ntiles);
  const int co0 = mt * TM;
  const int g0 = (tile - mt * a.ntiles) * GB;  // first column group of this block
  const int k0 = split * a.kchunk;
  const int k1 = min(k0 + a.kchunk, a.M);
  const int nK = RAGGED ? (k1 - k0 + KP - 1) / KP : (k1 - k0) / KP;

  // ---- descriptors: dy is linear in the pixel index; the activation base is shifted so every tap offset is >= 0 -------
  const unsigned dybytes = (unsigned)a.M * a.Cout * 2u;
  const unsigned abytes = (unsigned)a.N * a.Hi * a.Wi * a.Cin * 2u;
  const unsigned shift = (S2 || (!UPS && a.ks == 3)) ? (unsigned)(a.Wi + 1) * a.Cin * 2u : 0u;
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, dybytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB =
      __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(a.a) - shift), 0, abytes + shift, 0x00020000);

  // ---- per-lane DMA sources ----------------------------------------------------------------------------------------------
  // piece pb (0 .. 8*G-1): sub-tile pb / 8, pixel rows 4*(pb % 8) .. +3; lane -> row + lane/16, physical 16-B chunk lane%16
  const int cphys = lane & 15;
  unsigned voffA[NPA];
  int rowA[NPA];   // RAGGED: the piece's pixel row inside the K tile
#pragma unroll
  for (int p = 0; p < NPA; p++) {
    const int pb = wave * NPA + p;
    const int sub = pb >> 3, row = (pb & 7) * 4 + (lane >> 4);
    rowA[p] = (KH == 2 ? sub * 32 : 0) + row;
    const int clog = ((((cphys >> 1) ^ (((row & 3) << 1) | ((row >> 3) & 1))) << 1) | (cphys & 1)) * 8;
    // KH == 1: sub-tile = 128-channel group of the block's couts; KH == 2 (GA == 1): sub-tile = 32-pixel half of the K tile
    const int co = co0 + (KH == 2 ? 0 : sub * 128) + clog;
    voffA[p] = co < a.Cout ? (unsigned)(((KH == 2 ? sub * 32 : 0) + row) * a.Cout + co) * 2u : SENT;
  }
  unsigned voffB[NPB], voffBL[NPB], voffBR[NPB];  // plain / first pixel of a row masked / last pixel masked
  int kyB[NPB], kxB[NPB];                           // tap of the piece's column group (wave-uniform)
  unsigned tapoB[NPB];                              // wave-uniform byte offset of the tap (+ shift), non-upsampled case
  int rowB[NPB];
  const int my_nt0 = tile - mt * a.ntiles;
  const int hky = uni(my_nt0 / a.gpt), hhalf = my_nt0 - hky * a.gpt;  // HALO: kernel row and 128-channel slice of this block
#pragma unroll
  for (int p = 0; p < NPB; p++) {
    if constexpr (HALO) {
      const int q = p < FP ? wave + 8 * p : NQ - 1;     // piece: LDS rows 4 q .. 4 q + 3 <-> source pixels x0 - 1 + row
      const int row = q * 4 + (lane >> 4);
      const int clog = ((((cphys >> 1) ^ (((row & 3) << 1) | ((row >> 3) & 1))) << 1) | (cphys & 1)) * 8;
      const bool ok = row < KP + 2 && (p < FP || wave == 7);
      const unsigned v = (unsigned)(row * a.Cin + hhalf * 128 + clog) * 2u;
      kyB[p] = hky; kxB[p] = 0; rowB[p] = row;
      tapoB[p] = (unsigned)(hky * a.Wi) * a.Cin * 2u;
      voffB[p] = ok ? v : SENT;
      voffBL[p] = (ok && row != 0) ? v : SENT;
      voffBR[p] = (ok && row != KP + 1) ? v : SENT;
      continue;
    }
    const int pb = wave * NPB + p;
    const int sub = pb >> 3, row = (pb & 7) * 4 + (lane >> 4);
    const int clog = ((((cphys >> 1) ^ (((row & 3) << 1) | ((row >> 3) & 1))) << 1) | (cphys & 1)) * 8;
    const int g = g0 + sub;
    const int tap = uni(g / a.gpt), ci = (g - tap * a.gpt) * 128 + clog;
    const bool ok = g < a.ngroups;
    kyB[p] = S2 ? tap >> 2 : (a.ks == 3 ? uni(tap / 3) : 1);
    kxB[p] = S2 ? tap & 3 : (a.ks == 3 ? tap - (tap / 3) * 3 : 1);
    tapoB[p] = (S2 || (!UPS && a.ks == 3)) ? (unsigned)(kyB[p] * a.Wi + kxB[p]) * a.Cin * 2u : 0u;
    rowB[p] = row;
    const unsigned v = S2 ? (unsigned)(2 * row * a.Cin + ci) * 2u : (UPS ? (unsigned)ci * 2u : (unsigned)(row * a.Cin + ci) * 2u);
    voffB[p] = ok ? v : SENT;
    voffBL[p] = (ok && row != 0) ? v : SENT;
    voffBR[p] = (ok && row != 31) ? v : SENT;
  }

  // ---- fragment read addresses (bytes inside a slot): one per 16-channel block; the second half (+4 pixel rows) is an immediate ------------
  const int G = lane >> 4, rr = (lane & 15) >> 2, qq = lane & 3;
  const int fkey = (rr << 1) | (G & 1);  // swizzle key of pixel rows 8 G + rr and 8 G + rr + 4
  int aoff[BM], boff[BN], boff1[HALO ? BN : 1];
#pragma unroll
  for (int i = 0; i < BM; i++) {
    const int ch = wm * (TM / WM) + i * 16 + 4 * qq;  // channel inside the TM-wide operand
    const int sub = ch >> 7, c = ch & 127;
    aoff[i] = sub * SUB + (G * 8 + rr) * 256 + ((((c >> 4) & 7) ^ fkey) << 5) + (c & 15) * 2;
  }
#pragma unroll
  for (int j = 0; j < BN; j++) {
    const int ch = wn * (TN / WN) + j * 16 + 4 * qq;
    const int sub = ch >> 7, c = ch & 127;
    if constexpr (HALO) {  // tap kx = sub: pixel rows 8 G + rr + kx and + 4 of the halo tile (the + 4 may cross an 8-row boundary: its own swizzle key)
      const int r0 = G * 8 + rr + sub, r1 = r0 + 4;
      boff[j] = ASZ + r0 * 256 + ((((c >> 4) & 7) ^ (((r0 & 3) << 1) | ((r0 >> 3) & 1))) << 5) + (c & 15) * 2;
      boff1[j] = ASZ + r1 * 256 + ((((c >> 4) & 7) ^ (((r1 & 3) << 1) | ((r1 >> 3) & 1))) << 5) + (c & 15) * 2;
    } else {
      boff[j] = GA * SUB + sub * SUB + (G * 8 + rr) * 256 + ((((c >> 4) & 7) ^ fkey) << 5) + (c & 15) * 2;
    }
  }

  // Bias gradient (column sums of dy) rides along on the matrix pipe: dy fragment x all-ones fragment, in the wave whose N
  // index equals the cout block.  The ntiles blocks that share a (split, cout tile) take the K tiles round-robin, so every
  // block does 1/ntiles of it (one block doing all of it would set the critical path of a single-round launch) -- instead of
  // a second 2 B/elem pass over dy.
  const int my_nt = tile - mt * a.ntiles;
  const bool do_bias = a.bslab != nullptr && 2 * wn < BM;  // wave (wm, wn) sums cout blocks 2 wn and 2 wn + 1 of its row
  int bias_cnt = my_nt;
  f32x4 accb[2];
#pragma unroll
  for (int h = 0; h < 2; h++)
#pragma unroll
    for (int r = 0; r < 4; r++) accb[h][r] = 0.f;
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; e++) ones[e] = (bf16)1.0f;

  f32x4 acc[BM][BN];  // acc[i][j][r]: cout block i, row 4 * (lane >> 4) + r; column block j, column lane & 15
#pragma unroll
  for (int i = 0; i < BM; i++)
#pragma unroll
    for (int j = 0; j < BN; j++)
#pragma unroll
      for (int r = 0; r < 4; r++) acc[i][j][r] = 0.f;

  // ---- DMA issue state: tile `it` starts at pixel pt = (n, y, x0), all wave-uniform -----------------------------------------
  int it = 0;
  int pt = k0;
  int px0, py, pn;
  {
    const int hw = a.Ho * a.Wo;
    pn = uni(k0 / hw);
    const int r = k0 - pn * hw;
    py = uni(r / a.Wo);
    px0 = r - py * a.Wo;
  }
  auto issue = [&](int slot) {
    const bool live = it < nK;
    const unsigned soA = (unsigned)pt * a.Cout * 2u;
    const int left = a.M - pt;   // RAGGED: rows of this K tile that exist
#pragma unroll
    for (int p = 0; p < NPA; p++) {
      const int pb = wave * NPA + p;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, LPTR(smem + slot + pb * 1024), 16, (live && (!RAGGED || rowA[p] < left)) ? voffA[p] : SENT, soA, 0, 0);
    }
#pragma unroll
    for (int p = 0; p < NPB; p++) {
      if constexpr (HALO) {
        if (p == FP && wave != 7) continue;  // wave-uniform: the last piece (halo rows KP .. KP + 3) is wave 7's
        const int yy = py + kyB[p] - 1;
        const bool yok = (unsigned)yy < (unsigned)a.Ho;
        const unsigned v = p < FP ? (px0 == 0 ? voffBL[p] : voffB[p]) : (px0 + KP == a.Wo ? voffBR[p] : voffB[p]);
        const unsigned so = (unsigned)pt * a.Cin * 2u + tapoB[p];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, LPTR(smem + slot + ASZ + (p < FP ? wave + 8 * p : NQ - 1) * 1024), 16, (live && yok) ? v : SENT, yok ? so : 0u, 0, 0);
        continue;
      }
      const int pb = wave * NPB + p;
      const int yy = S2 ? 2 * py + kyB[p] - 1 : py + kyB[p] - 1;
      const bool yok = (unsigned)yy < (unsigned)(S2 ? a.Hi : a.Ho);
      const bool edgeL = kxB[p] == 0 && px0 == 0, edgeR = kxB[p] == (S2 ? 3 : 2) && px0 + 32 == a.Wo;
      unsigned v = edgeL ? voffBL[p] : (edgeR ? voffBR[p] : voffB[p]);
      unsigned so;
      if constexpr (S2) {
        so = (unsigned)((pn * a.Hi + 2 * py) * a.Wi + 2 * px0) * a.Cin * 2u + tapoB[p];
      } else if constexpr (UPS) {
        const int xx = px0 + rowB[p] + kxB[p] - 1;  // masked lanes never use it
        v = v == SENT ? SENT : v + (unsigned)(xx >> 1) * a.Cin * 2u;
        so = (unsigned)((pn * a.Hi + (yy >> 1)) * a.Wi) * a.Cin * 2u;
      } else {
        so = (unsigned)pt * a.Cin * 2u + tapoB[p];
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, LPTR(smem + slot + ASZ + pb * 1024), 16, (live && yok && (!RAGGED || rowB[p] < left)) ? v : SENT, yok ? so : 0u, 0, 0);
    }
    it++;
    pt += KP;
    px0 += KP;
    if (px0 == a.Wo) {
      px0 = 0;
      if (++py == a.Ho) { py = 0; pn++; }
    }
  };

  auto wait_ring = [&]() {  // all but the newest PF - 1 tiles' pieces of this wave have landed
    constexpr int AH = PF - 1;
    if constexpr (HALO) {
      if (wave == 7) wait_vmcnt<AH * NP>(); else wait_vmcnt<AH * (NP - 1)>();
    } else {
      wait_vmcnt<AH * NP>();
    }
  };
#pragma unroll
  for (int u = 0; u < PF; u++) issue(u * SLOT);
  wait_ring();
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_setprio(1);
  if (grp == 1) __builtin_amdgcn_s_barrier();

  union Frag { bf16x8 v; s16x4 h[2]; };
  Frag af[KH][BM], bfr[KH][BN];
  int slot_rd = 0, slot_wr = PF * SLOT;
#pragma unroll 1
  for (int t = 0; t < nK; t++) {
    const unsigned sb = (unsigned)(size_t)LPTR(smem) + (unsigned)slot_rd;
#pragma unroll
    for (int j = 0; j < BN; j++) {
      bfr[0][j].h[0] = tr_read<0>(sb + boff[j]);
      bfr[0][j].h[1] = HALO ? tr_read<0>(sb + boff1[HALO ? j : 0]) : tr_read<1024>(sb + boff[j]);
    }
#pragma unroll
    for (int i = 0; i < BM; i++) {
      af[0][i].h[0] = tr_read<0>(sb + aoff[i]);
      af[0][i].h[1] = tr_read<1024>(sb + aoff[i]);
    }
    if constexpr (KH == 2) {  // second 32-pixel half: 32 halo rows / one dy sub-tile further on (the swizzle key repeats every 16 rows)
#pragma unroll
      for (int j = 0; j < BN; j++) {
        bfr[KH - 1][j].h[0] = tr_read<8192>(sb + boff[j]);
        bfr[KH - 1][j].h[1] = tr_read<8192>(sb + boff1[HALO ? j : 0]);
      }
#pragma unroll
      for (int i = 0; i < BM; i++) {
        af[KH - 1][i].h[0] = tr_read<SUB>(sb + aoff[i]);
        af[KH - 1][i].h[1] = tr_read<SUB + 1024>(sb + aoff[i]);
      }
    }
    issue(slot_wr);
    slot_rd = slot_rd + SLOT == NBUF * SLOT ? 0 : slot_rd + SLOT;
    slot_wr = slot_wr + SLOT == NBUF * SLOT ? 0 : slot_wr + SLOT;
    wait_ring();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int h = 0; h < KH; h++)
#pragma unroll
    for (int i = 0; i < BM; i++)
#pragma unroll
      for (int j = 0; j < BN; j++) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(af[h][i].v), "v"(bfr[h][j].v));
      }
    const bool bias_now = do_bias && bias_cnt == 0;
    bias_cnt = bias_cnt == 0 ? a.ntiles - 1 : bias_cnt - 1;
    if (bias_now) {
#pragma unroll
      for (int h = 0; h < KH; h++)
#pragma unroll
      for (int i = 0; i < BM; i++)
        if ((i >> 1) == wn)  // accumulator pinned to VGPRs ("+v"): the AGPRs of the main accumulators are left exactly as they are
          // s_nop: the compiler rematerialises `ones` with v_mov right before the statement and pads nothing for inline asm
          asm volatile("s_nop 3\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(accb[i & 1]) : "v"(af[h][i].v), "v"(ones));
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();
  wait_vmcnt<0>();
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");

  if (do_bias && (lane & 15) == 0) {  // every column of accb holds the same sums: column 0 lives in lanes 0, 16, 32, 48
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int co = co0 + wm * (TM / WM) + (2 * wn + h) * 16 + 4 * G + r;
        if (co < a.Cout) a.bslab[((size_t)split * a.ntiles + my_nt) * a.Cout + co] = accb[h][r];
      }
  }
  // ---- slab store: lane owns column (l & 15) of each N block, 4 couts per accumulator ---------------------------------------------
  float* slab = a.slab + (size_t)split * a.Cout * T * a.Cin;
#pragma unroll
  for (int j = 0; j < BN; j++) {
    const int nn = wn * (TN / WN) + j * 16 + (lane & 15);  // column inside the block's TN
    const int g = g0 + (nn >> 7);
    if (!HALO && g >= a.ngroups) continue;
    const int tap = HALO ? hky * 3 + (nn >> 7) : g / a.gpt, ci = HALO ? hhalf * 128 + (nn & 127) : (g - tap * a.gpt) * 128 + (nn & 127);
#pragma unroll
    for (int i = 0; i < BM; i++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int co = co0 + wm * (TM / WM) + i * 16 + 4 * G + r;
        if (co < a.Cout) slab[((size_t)co * T + tap) * a.Cin + ci] = acc[i][j][r];
      }
  }
#endif
}

template <int GA, int GB, int WM, int WN, bool S2 = false, bool HALO = false, bool UPS = false, int KP = 32>
__global__ __launch_bounds__(512) void wgrad_pp_kernel(Args a) {
  wgrad_pp_body<GA, GB, WM, WN, S2, HALO, UPS, KP, false>(a, blockIdx.x, gridDim.x);
}

// Grouped form: ONE launch for a table of independent Linear weight gradients dW_p [Cout_p][Cin_p] = dY_p^T . X_p (ks = 1), every problem unsplit (its whole
// reduction in one block per 256 x 256 output tile, written straight to its destination: no slabs, no reduce launch) -- for call sites that hold many of them at
// once: the four Linears x 28 blocks of LightningDiT's backward pass (7700 tiles = 30 full rounds of the chip, where one weight gradient at batch 16 is 25-120 tiles
// and needed a 2- to 8-way split-K plus a slab reduce to fill it), the four Linears of a ViT block.  Entry p owns the flat blocks [start_p, start_p + mtiles_p * ntiles_p).
struct GEntry { Args a; unsigned start, blocks; };
template <bool RAGGED>
__global__ __launch_bounds__(512) void wgrad_pp_grouped_kernel(const GEntry* __restrict__ tab, int n) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {   // last entry whose start <= blockIdx.x (block-uniform: scalar loads)
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].start <= blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const int p = __builtin_amdgcn_readfirstlane(lo);
  // every problem starts at a multiple of 8 flat blocks (dmvae_linear_wgrad_grouped_fill), so that (flat - start) & 7 is still the XCD the hardware put this block on
  // and xcd_remap keeps a problem's neighbouring tiles -- the ones that share a dY row panel -- on one XCD's L2; the up to 7 padding blocks behind a problem leave here
  const unsigned local = blockIdx.x - tab[p].start;
  if (local >= tab[p].blocks) return;
  const Args a = tab[p].a;
  wgrad_pp_body<2, 2, 2, 4, false, false, false, 32, RAGGED>(a, local, tab[p].blocks);
}
// The same launch with the tiles PLACED: a problem's tiles are cut into chunks of whole cout-tile rows of about 32 tiles (one round of an XCD's 32 CUs) and every
// chunk is given to ONE XCD (dmvae_linear_wgrad_grouped_plan: longest chunk first to the least loaded XCD); flat block f runs position f >> 3 of XCD f & 7's list
// (the hardware deals consecutive workgroups round-robin over the XCDs).  The 32 tiles an XCD runs at a time are then rows_per_chunk x ntiles tiles of ONE problem:
// rows + ntiles operand panels in that XCD's L2 for 32 tiles.  The first form spread every problem over all eight XCDs (an eighth of its tiles each, nine of
// LightningDiT's qkv gradient's 70): every x panel fetched by eight L2s, two or three thin problems sharing an XCD -- 16 GB read per launch at batch 16 where
// the operands are 4.2 GB, at 4.9 TB/s of fabric traffic: bandwidth-bound at 1.0 PFLOP/s (profiles/r5_dmd_stage_traffic_pmc.txt).
struct GChunk { unsigned entry, tile0, start, blocks; };
struct GPlan { unsigned xoff[9]; };
template <bool RAGGED>
__global__ __launch_bounds__(512) void wgrad_pp_grouped_xcd_kernel(const GEntry* __restrict__ tab, const GChunk* __restrict__ ch, GPlan plan) {
  const unsigned x = blockIdx.x & 7u, pos = blockIdx.x >> 3;
  int lo = (int)plan.xoff[x], hi = (int)plan.xoff[x + 1] - 1;
  if (hi < lo) return;
  while (lo < hi) {   // last chunk of this XCD's list whose start <= pos (block-uniform: scalar loads)
    const int mid = (lo + hi + 1) >> 1;
    if (ch[mid].start <= pos) lo = mid; else hi = mid - 1;
  }
  const int c = __builtin_amdgcn_readfirstlane(lo);
  const unsigned local = pos - ch[c].start;
  if (local >= ch[c].blocks) return;      // behind the XCD's last chunk
  const Args a = tab[ch[c].entry].a;
  wgrad_pp_body<2, 2, 2, 4, false, false, false, 32, RAGGED>(a, ch[c].tile0 + local, 0u);
}
// bias gradients of the grouped launch: db_p[c] = sum over the problem's ntiles_p partial rows (the bias sums ride the matrix pipe round-robin over a cout tile's blocks)
struct GBias { const float* part; float* out; int nparts, C; unsigned start; };
__global__ __launch_bounds__(256) void wgrad_grouped_bias_kernel(const GBias* __restrict__ tab, int n) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].start <= blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const GBias e = tab[lo];
  const int c = (int)(blockIdx.x - e.start) * 256 + threadIdx.x;
  if (c >= e.C) return;
  float s = 0.f;
  for (int k = 0; k < e.nparts; k++) s += e.part[(size_t)k * e.C + c];
  e.out[c] = s;
}

template <int GA, int GB, int WM, int WN, bool S2 = false, bool HALO = false, bool UPS = false, int KP = 32>
int launch(const Args& a, int splits, hipStream_t st) {
  constexpr int lds = (HALO ? (KP == 64 ? 4 : WG_NBUF) : WG_NBUF) * (GA * KP + (HALO ? KP + 4 : GB * 32)) * 256;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_pp_kernel<GA, GB, WM, WN, S2, HALO, UPS, KP>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done = true;
  }
  hipLaunchKernelGGL((wgrad_pp_kernel<GA, GB, WM, WN, S2, HALO, UPS, KP>), dim3((unsigned)(splits * a.mtiles * a.ntiles)), dim3(512), lds, st, a);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

}  // namespace dmvae_wgrad_pp

// The 128 x 384 tile's halo form with 64-pixel K tiles: 3x3 stride 1, output rows of a multiple of 64 pixels (a K tile never straddles a row).
static constexpr bool wgrad_pp_halo_on() { return true; }
static bool wgrad_pp_k64(const dmvae_conv_desc* d, int cfg) {
  constexpr bool on = true;
  return on && wgrad_pp_halo_on() && cfg == 1 && d->ks == 3 && d->stride <= 1 && !d->upsample && !d->transposed && d->w % 64 == 0;
}

// Plan shared by the workspace query and the launch: returns 0 when the ping-pong kernel does not cover the shape.
int dmvae_wgrad_pp_plan(const dmvae_conv_desc* d, int* splits_out, int* kchunk_out, int* cfg_out) {
  const bool s2 = d->ks == 4 && d->stride == 2 && !d->upsample && !d->transposed && d->h % 2 == 0 && d->w % 2 == 0;  // 4x4 stride 2: its own instantiation
  if (!s2 && (d->stride == 2 || d->upsample == 2 || d->ks == 4 || d->transposed)) return 0;  // other strided / 4x4 gathers: the general kernel (conv_wgrad.hip)
  const int ups = d->upsample ? 1 : 0;
  const int wo = s2 ? d->w / 2 : (ups ? 2 * d->w : d->w);
  const long long M = s2 ? (long long)d->n * (d->h / 2) * (d->w / 2) : (long long)d->n * d->h * d->w * (ups ? 4 : 1);
  // Cout = 64 on the 1x1 form (the PatchGAN's first layer on its im2col: functional.ConvK4Fn): rows past Cout are masked, so a quarter-full 256-row tile serves it --
  // the launch is bound by reading the two operands once, and a zero-padded 128-channel copy of dY (fill + copy + twice the read) cost more than the idle rows
  const bool c64 = d->ks == 1 && d->cout == 64;
  if (d->cin % 128 != 0 || (d->cout % 128 != 0 && !c64) || wo % 32 != 0 || M < 4096) return 0;
  if ((long long)M * d->cout * 2 >= (1ll << 31) || (long long)d->n * d->h * d->w * d->cin * 2 + (1ll << 22) >= (1ll << 31)) return 0;
  const int T = d->ks * d->ks;
  const int ngroups = T * (d->cin / 128);
  // Tile configuration: 256 x 256 (two 128-channel column groups) or 128 x 384 (three).  Ragged edges are masked (rows past Cout load zeros and are not
  // stored, a column group past the last is skipped), so either fits any shape; what differs is the padding each wastes and the tile's own efficiency --
  // the 128 x 384 tile stages 96 instead of 128 FLOP per LDS byte and runs 24 instead of 32 MFMAs between barriers (measured ~0.78 of the larger tile's rate).
  int cfg, mtiles, ntiles;
  {
    const int m0 = (d->cout + 255) / 256, n0 = (ngroups + 1) / 2, m1 = d->cout / 128, n1 = (ngroups + 2) / 3;
    const double useful = (double)d->cout * ngroups * 128;
    const double e0 = useful / ((double)m0 * 256 * n0 * 256), e1 = 0.78 * useful / ((double)m1 * 128 * n1 * 384);
    constexpr bool relax = true;
    const bool exact0 = d->cout % 256 == 0 && (d->cin / 128) % 2 == 0;
    // The 128 x 384 tile's halo form with 64-pixel K tiles (190 instead of 128 FLOP per staged byte) also beats the 256 x 256 tile on shapes that tile fits
    // exactly once the reduction is long: +3-8 % from 2^19 pixels on, +-0 below (DESIGN_HISTORY.md 8.13).
    constexpr int force = -1;
    const bool halo_ok = d->ks == 3 && d->stride <= 1 && !d->upsample && !d->transposed && d->w % 64 == 0 && wgrad_pp_halo_on();
    if (c64) { cfg = 0; mtiles = m0; ntiles = n0; }
    else if (halo_ok && (force == 1 || (force < 0 && M >= (1ll << 19)))) { cfg = 1; mtiles = m1; ntiles = n1; }
    else if (exact0 || (relax && e0 > e1)) { cfg = 0; mtiles = m0; ntiles = n0; }
    else { cfg = 1; mtiles = m1; ntiles = n1; }
  }
  const int tiles = mtiles * ntiles;
  // fill the 256 CUs with whole blocks (1 block per CU): the fewest rounds that still leave >= 16 K tiles per split
  const int ktiles = (int)(M / 32);
  int best = 1;
  double best_eff = 0;
  // (at least two rounds of smaller blocks, for launches next to an overlapped collective's resident kernel, was measured and not adopted: DESIGN_HISTORY.md 8.6)
  constexpr int min_rounds = 1;
  for (int s = 1; s <= ktiles / 16 && s * tiles <= 4096; s++) {
    const int blocks = s * tiles;
    const int rounds = (blocks + 255) / 256;
    if (rounds < min_rounds && (s + 1) <= ktiles / 16 && (s + 1) * tiles <= 4096) continue;
    const int kt = (ktiles + s - 1) / s;  // K tiles per split (critical path per block)
    const double eff = (double)ktiles * tiles / ((double)rounds * 256 * kt) / (1.0 + 24.0 / kt);  // useful / provisioned, with a fixed per-block cost
    if (eff > best_eff * 1.0001) { best_eff = eff; best = s; }
  }
  int kt = (ktiles + best - 1) / best;
  if (wgrad_pp_k64(d, cfg)) kt = (kt + 1) & ~1;   // whole 64-pixel K tiles per split
  *kchunk_out = kt * 32;
  *splits_out = (ktiles + kt - 1) / kt;
  *cfg_out = cfg;
  return 1;
}

int dmvae_wgrad_pp_launch(const void* dy, const void* act, float* slab, float* bslab, const dmvae_conv_desc* d, int splits, int kchunk,
                          int cfg, hipStream_t stream) {
  using namespace dmvae_wgrad_pp;
  Args a;
  a.dy = (const bf16*)dy; a.a = (const bf16*)act; a.slab = slab; a.bslab = bslab;
  a.N = d->n; a.Hi = d->h; a.Wi = d->w; a.Cin = d->cin; a.Cout = d->cout; a.ks = d->ks;
  a.ups = d->upsample ? 1 : 0;
  const bool s2 = d->ks == 4 && d->stride == 2;
  a.Ho = s2 ? d->h / 2 : (a.ups ? 2 * d->h : d->h); a.Wo = s2 ? d->w / 2 : (a.ups ? 2 * d->w : d->w);
  a.M = a.N * a.Ho * a.Wo;
  a.kchunk = kchunk;
  a.gpt = d->cin / 128;
  a.ngroups = d->ks * d->ks * a.gpt;
  if (cfg == 0) {
    a.mtiles = (d->cout + 255) / 256; a.ntiles = (a.ngroups + 1) / 2;
    if (a.ups) return launch<2, 2, 2, 4, false, false, true>(a, splits, stream);
    return s2 ? launch<2, 2, 2, 4, true>(a, splits, stream) : launch<2, 2, 2, 4>(a, splits, stream);
  }
  a.mtiles = d->cout / 128; a.ntiles = (a.ngroups + 2) / 3;
  // 3x3 stride 1: the halo form (three taps of one kernel row per block; ntiles = 3 * gpt either way)
  if (wgrad_pp_k64(d, cfg) && kchunk % 64 == 0) return launch<1, 3, 2, 4, false, true, false, 64>(a, splits, stream);
  if (wgrad_pp_halo_on() && !s2 && d->ks == 3 && !a.ups) return launch<1, 3, 2, 4, false, true>(a, splits, stream);
  if (a.ups) return launch<1, 3, 2, 4, false, false, true>(a, splits, stream);
  return s2 ? launch<1, 3, 2, 4, true>(a, splits, stream) : launch<1, 3, 2, 4>(a, splits, stream);
}


// ---- grouped Linear weight gradients (see wgrad_pp_grouped_kernel) ---------------------------------------------------------------------------------
extern "C" size_t dmvae_linear_wgrad_grouped_entry_bytes(void) { return sizeof(dmvae_wgrad_pp::GEntry); }
extern "C" size_t dmvae_linear_wgrad_grouped_bias_entry_bytes(void) { return sizeof(dmvae_wgrad_pp::GBias); }
extern "C" int dmvae_linear_wgrad_grouped_supported(int M, int cout, int cin) {
  return (M >= 32 && cout >= 128 && cin >= 128 && cout % 128 == 0 && cin % 128 == 0 && (long long)M * cout * 2 < (1ll << 31) && (long long)M * cin * 2 + (1ll << 22) < (1ll << 31)) ? 1 : 0;
}
// Fill host-side table records for one problem: dy [M][cout], x [M][cin] bf16 row-major, dw f32 [cout][cin] (written, not accumulated), bias_part f32 scratch of
// dmvae_linear_wgrad_grouped_bias_parts(cin) * cout floats and db f32 [cout] (both NULL: no bias gradient).  *start is advanced by the problem's block count, *bias_start
// by its bias blocks.  Returns 0, or -22 on a shape the kernel does not take.
extern "C" int dmvae_linear_wgrad_grouped_bias_parts(int cin) { return ((cin / 128) + 1) / 2; }
extern "C" int dmvae_linear_wgrad_grouped_fill(void* entry, void* bias_entry, const void* dy, const void* x, void* dw, void* bias_part, void* db, int M, int cout, int cin,
                                               unsigned* start, unsigned* bias_start) {
  using namespace dmvae_wgrad_pp;
  DMVAE_CHECK_ARG(entry && dy && x && dw && start && bias_start && (!db || (bias_part && bias_entry)), "linear_wgrad_grouped_fill: null pointer");
  DMVAE_CHECK_ARG(dmvae_linear_wgrad_grouped_supported(M, cout, cin), "linear_wgrad_grouped_fill: M=%d cout=%d cin=%d (M >= 32, cout / cin multiples of 128, operands < 2 GiB)", M, cout, cin);
  GEntry g;
  Args& a = g.a;
  a.dy = (const bf16*)dy; a.a = (const bf16*)x; a.slab = (float*)dw; a.bslab = db ? (float*)bias_part : nullptr;
  a.N = 1; a.Hi = 1; a.Wi = M; a.Cin = cin; a.Ho = 1; a.Wo = M; a.Cout = cout; a.ks = 1; a.ups = 0; a.M = M;
  a.kchunk = (M + 31) / 32 * 32;
  a.gpt = cin / 128; a.ngroups = a.gpt;
  a.mtiles = (cout + 255) / 256; a.ntiles = (a.ngroups + 1) / 2;
  g.blocks = (unsigned)(a.mtiles * a.ntiles);
  g.start = *start;                                  // a multiple of 8: see wgrad_pp_grouped_kernel
  *start += (g.blocks + 7u) & ~7u;
  *reinterpret_cast<GEntry*>(entry) = g;
  if (db) {
    GBias b{(const float*)bias_part, (float*)db, a.ntiles, cout, *bias_start};
    *bias_start += (unsigned)((cout + 255) / 256);
    *reinterpret_cast<GBias*>(bias_entry) = b;
  }
  return 0;
}
// Placement of a filled table's tiles on the XCDs (wgrad_pp_grouped_xcd_kernel): chunks_out receives up to max_chunks records of dmvae_linear_wgrad_grouped_chunk_bytes(),
// grouped by XCD (xoff[x] .. xoff[x + 1]), *grid the launch's block count.  Host-only; returns 0, or -22 when max_chunks is too small.
extern "C" size_t dmvae_linear_wgrad_grouped_chunk_bytes(void) { return sizeof(dmvae_wgrad_pp::GChunk); }
extern "C" int dmvae_linear_wgrad_grouped_plan(const void* table, int n, void* chunks_out, int max_chunks, int* n_chunks, unsigned* xoff, unsigned* grid) {
  using namespace dmvae_wgrad_pp;
  DMVAE_CHECK_ARG(table && n > 0 && chunks_out && n_chunks && xoff && grid, "linear_wgrad_grouped_plan: null pointer");
  const GEntry* tab = (const GEntry*)table;
  struct C { unsigned entry, tile0, blocks; int xcd; };
  std::vector<C> cs;
  for (int p = 0; p < n; p++) {
    const int nt = tab[p].a.ntiles, mt = tab[p].a.mtiles;
    const int rows = std::max(1, (32 + nt / 2) / nt);            // whole cout-tile rows of about 32 tiles
    for (int r0 = 0; r0 < mt; r0 += rows) cs.push_back(C{(unsigned)p, (unsigned)(r0 * nt), (unsigned)(std::min(rows, mt - r0) * nt), 0});
  }
  DMVAE_CHECK_ARG((int)cs.size() <= max_chunks, "linear_wgrad_grouped_plan: %d chunks, room for %d", (int)cs.size(), max_chunks);
  std::vector<int> order(cs.size());
  for (size_t i = 0; i < cs.size(); i++) order[i] = (int)i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cs[a].blocks > cs[b].blocks; });
  unsigned load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i : order) {                                            // longest first to the least loaded XCD (lowest index on ties: deterministic)
    int best = 0;
    for (int x = 1; x < 8; x++) if (load[x] < load[best]) best = x;
    cs[i].xcd = best;
    load[best] += cs[i].blocks;
  }
  GChunk* out = (GChunk*)chunks_out;
  unsigned k = 0, longest = 0;
  for (int x = 0; x < 8; x++) {                                    // problem order inside an XCD's list
    xoff[x] = k;
    unsigned pos = 0;
    for (size_t i = 0; i < cs.size(); i++)
      if (cs[i].xcd == x) { out[k++] = GChunk{cs[i].entry, cs[i].tile0, pos, cs[i].blocks}; pos += cs[i].blocks; }
    longest = std::max(longest, pos);
  }
  xoff[8] = k;
  *n_chunks = (int)k;
  *grid = longest * 8u;
  return 0;
}
extern "C" int dmvae_linear_wgrad_grouped_xcd(const void* table, const void* chunks, const unsigned* xoff, unsigned grid, int ragged, const void* bias_table, int n_bias,
                                              unsigned bias_blocks, hipStream_t stream) {
  using namespace dmvae_wgrad_pp;
  DMVAE_CHECK_ARG(table && chunks && xoff && grid > 0, "linear_wgrad_grouped_xcd: empty plan");
  constexpr int lds = WG_NBUF * (2 * 32 + 2 * 32) * 256;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_pp_grouped_xcd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_pp_grouped_xcd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done = true;
  }
  GPlan plan;
  for (int i = 0; i < 9; i++) plan.xoff[i] = xoff[i];
  if (ragged) hipLaunchKernelGGL(wgrad_pp_grouped_xcd_kernel<true>, dim3(grid), dim3(512), lds, stream, (const GEntry*)table, (const GChunk*)chunks, plan);
  else hipLaunchKernelGGL(wgrad_pp_grouped_xcd_kernel<false>, dim3(grid), dim3(512), lds, stream, (const GEntry*)table, (const GChunk*)chunks, plan);
  DMVAE_CHECK_LAUNCH();
  if (bias_table && n_bias > 0 && bias_blocks > 0) {
    hipLaunchKernelGGL(wgrad_grouped_bias_kernel, dim3(bias_blocks), dim3(256), 0, stream, (const GBias*)bias_table, n_bias);
    DMVAE_CHECK_LAUNCH();
  }
  return 0;
}
extern "C" int dmvae_linear_wgrad_grouped(const void* table, int n, unsigned total_blocks, int ragged, const void* bias_table, int n_bias, unsigned bias_blocks,
                                          hipStream_t stream) {
  using namespace dmvae_wgrad_pp;
  DMVAE_CHECK_ARG(table && n > 0 && total_blocks > 0, "linear_wgrad_grouped: empty table");
  constexpr int lds = WG_NBUF * (2 * 32 + 2 * 32) * 256;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_pp_grouped_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_pp_grouped_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done = true;
  }
  if (ragged) hipLaunchKernelGGL(wgrad_pp_grouped_kernel<true>, dim3(total_blocks), dim3(512), lds, stream, (const GEntry*)table, n);
  else hipLaunchKernelGGL(wgrad_pp_grouped_kernel<false>, dim3(total_blocks), dim3(512), lds, stream, (const GEntry*)table, n);
  DMVAE_CHECK_LAUNCH();
  if (bias_table && n_bias > 0 && bias_blocks > 0) {
    hipLaunchKernelGGL(wgrad_grouped_bias_kernel, dim3(bias_blocks), dim3(256), 0, stream, (const GBias*)bias_table, n_bias);
    DMVAE_CHECK_LAUNCH();
  }
  return 0;
}
