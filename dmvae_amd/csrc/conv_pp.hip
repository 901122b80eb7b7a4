// Implicit-GEMM convolution (forward and, with flipped weights, dgrad) for NHWC bf16 on gfx950 -- "ping-pong" kernel.
//
// Same contract as conv_fwd.hip (nn.Conv2d at models/flux_ae.py:32-35,63,65,67,101,237,274; nn.Linear at
// models/vae.py:58-62), restructured around what the PMC counters of the first kernel showed (profiles/
// r1_conv_fwd_v1_pmc.txt): 17 non-MFMA instructions per MFMA (64-bit im2col address math every K step), one K step of
// prefetch, and 64 FLOP per staged byte.  Here:
//
//   * tile 256(cout) x 256(pixels) or 128 x 512, 8 waves, each wave 128x64 or 64x128 as 32 accumulators of 16x16 (v_mfma_f32_16x16x32_bf16):
//     102-128 FLOP per byte staged through LDS instead of 64;
//   * K tile = 32 channels of one tap (64-B LDS rows, XOR-swizzled 16-B chunks, conflict-free ds_read_b128), NBUF-deep
//     LDS ring filled by LDS-DMA (buffer_load_dwordx4 ... lds) NBUF-1 tiles ahead, counted s_waitcnt vmcnt(N) -- the
//     queue is never drained inside the loop;
//   * im2col addressing is a per-lane 32-bit voffset fixed for the whole kernel (per tap: one select against a
//     precomputed 9-bit validity mask) plus a wave-uniform soffset; zero padding and ragged edges come from the
//     buffer descriptor's out-of-range rule (returns 0), verified on hardware by tools/probes/probe_buflds.hip;
//   * the two waves that share a SIMD (w and w+4) alternate roles every interval: one issues its 12 ds_read_b128 +
//     LDS-DMA while the other runs its 32 MFMAs (512 cycles) under s_setprio(1); two s_barrier per K tile keep the roles in step.
//
//   * plain 3x3 convs with a bf16 result run the HALO instantiations (described at the kernel template): the three kx taps of a (channel chunk, ky) read one
//     staged halo of the pixel tile, the weights come K-tile-major (dmvae_conv_desc.w_layout = 1) so that a weight tile is whole 128-B lines; tiles 256 x 256,
//     128 x 512 and, for 64 output channels, 64 x 1024.  What that was worth, and the timing experiments behind it: DESIGN_HISTORY.md 8.12 / 8.13.
//
// Hazards (B_k = k-th workgroup barrier; group 0 = waves 0-3, group 1 = waves 4-7, one barrier behind):
//   RAW  tile t+1 is read after B_{2t+2}; every wave waits (vmcnt) for its own pieces of t+1 at the end of its LOAD(t),
//        i.e. before B_{2t+1} (group 0) / B_{2t+2} (group 1).
//   WAR  the slot of tile t-1 is re-filled by DMA issued after B_{2t}; its last ds_reads (group 1, LOAD(t-1)) are
//        retired by lgkmcnt(0) before B_{2t}.
#include "common.h"
#include "dmvae_hip.h"
#include <cstdlib>
#include <atomic>
#include <mutex>
#include <type_traits>
#include <unordered_map>

// What was measured on this kernel and not adopted (each a source variant at the time, bit-identical where it computed the same thing) is kept as text, not as
// code: DESIGN_HISTORY.md 8.12 / 8.13 (the timing experiments behind the kx-halo form), 9.3 (LDS-staged against direct epilogue), 9.6 (one continuous K-tile stream per block,
// staggered XCD starts, cache-policy bits of the epilogue's stores and of the LDS-DMA: nt stores, plain DMA).

namespace dmvae_conv_pp {

struct Args {
  const bf16* x;      // [N, Hi, Wi, Cin]
  const bf16* w;      // [Cout, T, Cin]
  const float* bias;  // [Cout] or null
  const bf16* res;    // [N, Ho, Wo, Cout] or null
  void* y;            // [N, Ho, Wo, Cout] bf16 or f32
  int N, Hi, Wi, Cin, Ho, Wo, Cout;
  int ks, act, M, ctiles, total;  // total = pixel tiles x cout tiles
  int so, pd, sd;                 // gather geometry (dmvae_conv_geometry): tap k of output o reads source (o * so - pd + k) / sd when that is an in-range integer
  int Ml;                         // SUB: source pixels N * Hi * Wi (= output pixels of one parity class); M is set to the same value
  unsigned wsRow, wsTap, wsChunk; // byte strides of the weight operand per cout row / tap / 32-channel chunk (dmvae_conv_desc.w_layout)
  float* gnpart;                  // STATS: [pixel tile][wave column 0..3][Cout / 4][2] per-tile (sum, sum of squares) of the bf16 results, 4 channels each
  unsigned* sched;                // DYN: this stream's scheduling words -- [0..7] tiles claimed past the static first round, per XCD range; [8] blocks finished;
                                  // [16 + b] the tile block b runs next.  All zero between launches (the last block to finish resets them).
  unsigned long long* dbg;  // optional per-block s_memtime stamps (dmvae_debug_timing), null in production
};

constexpr unsigned SENT = 0x80000000u;  // voffset beyond any descriptor's num_records -> the DMA writes zeros

// 64-B rows, 4 rows per 256-B bank row.  Fragments are read for v_mfma_f32_16x16x32_bf16: lane l takes the 16-B chunk l >> 4 of row (l & 15), and
// ds_read_b128 serves the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... in one pass each; with the chunk XOR-ed by (-(row >> 2)) & 3 the four lanes of
// a group that share row % 4 land in four different 16-B slots of the bank row (derivation in DESIGN_HISTORY.md 3.1).
__device__ __forceinline__ int swz64(int row) { return (0 - (row >> 2)) & 3; }
__device__ __forceinline__ int hswz(int row) { return (row >> 1) & 2; }   // HALO rows: the key that stays conflict-free under a shift of 0..2 rows

// q = m / d, r = m % d for 0 <= m < 2^24 (exact in f32) via one reciprocal and a +-1 fix-up.  The host declines shapes with 2^24 pixels or more: a plain
// integer division as the other arm kept its reciprocal sequences alive (and spilled) across the whole kernel.
__device__ __forceinline__ void divmod_small(int m, int d, float inv_d, bool, int& q, int& r) {
  q = (int)((float)m * inv_d);
  r = m - q * d;
  if (r < 0) { q--; r += d; }
  if (r >= d) { q++; r -= d; }
}

// s_waitcnt vmcnt(N) through the builtin, not inline asm: the compiler's own wait-count pass then SEES the wait.  With the asm form it kept a VMEM event from
// before the K loop pending on a fragment register for ever (the loop's own LDS-DMA instructions make its count imprecise) and put an s_waitcnt vmcnt(0) in
// front of the second ds_read of every K tile.  Removing that drain changed nothing measurable (342 vs 340 us per launch in the step): by then the pieces
// of the next tiles have landed anyway -- the loop is not waiting on the DMA queue.
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  __builtin_amdgcn_s_waitcnt((N & 15) | 0x0F70 | ((N >> 4) << 14));   // gfx9 encoding: vmcnt[3:0] | expcnt 7 | lgkmcnt 15 | vmcnt[5:4] << 14
  asm volatile("" ::: "memory");
}

// GEN: the general gather geometry (4x4 taps, output stride 2, zero-insertion sources) -- a separate instantiation, because its per-tile setup and per-K-tile
// tap arithmetic cost the plain 3x3 / 1x1 kernel 13-19 % when compiled into it (measured in the step: 337 -> 382 us per launch).
// SUB: the transposed 4x4 stride-2 padding-1 conv (input gradient of the PatchGAN stride-2 convs, models/patchgan.py:125-133; and the FORWARD of Upsample's
// conv in its sub-pixel form, flux_ae.py:103-107 / dmvae_subpixel_weight) decomposed by output parity: output pixel (2y + py, 2x + px) reads exactly the 2x2
// source pixels (y - 1 + py + a, x - 1 + px + b) through taps (py + 2a, px + 2b) of the packed 16-tap operand -- four dense 2x2 convolutions writing
// interleaved pixels.  As a zero-insertion gather (GEN, sd = 2) twelve of the sixteen taps of every output pixel are masked: 4x the MFMA work.  A pixel
// tile holds output pixels of ONE parity class (wave-uniform tap set); the four classes of a source region are adjacent in the block order so that the
// region is fetched into the XCD's L2 once.
// DYN: tiles after a block's first are claimed from per-XCD counters instead of a static stride.  With the static stride a persistent block that starts
// late -- its CU held by another stream's kernel, e.g. an RCCL all-reduce overlapping backward: a block needs 128-160 KB of LDS, so it cannot share a CU
// with anything -- still owes all of its tiles and the launch ends a whole block-time late; with the counters the blocks that run take the tiles and the
// launch degrades by the fraction of CUs taken.  A block claims from its own XCD's contiguous range first (the L2 locality of the static order) and from
// the others' once that is exhausted.  Results do not depend on who computes which tile.
// STATS: the epilogue also sums the (bf16-rounded) results and their squares per 4 output channels and pixel tile -- the statistics pass of the GroupNorm
// that follows most decoder convs (flux_ae.py:62,64,71-76) then has nothing left to read: a finishing kernel combines the partials per (image, group) in
// f64 (groupnorm.hip::stats_from_quads_kernel).  Its own instantiation: the plain kernel's code is unchanged.
// HALO (plain 3x3, stride 1): the three kx taps of a (channel chunk, ky) read the SAME pixels shifted by one, so the activation operand of three consecutive
// K tiles is staged once, as the TP + 2 flat pixels m0 - 1 .. m0 + TP of source row offset ky - 1 (TP / 16 + 1 pieces instead of 3 * TP / 16), and tap kx
// reads pixel p's fragment from halo row p + kx.  A timing experiment (the same instruction stream with two of three activation pieces
// masked) measured +12-14 % on the decoder's shapes: what the L2 -> LDS staging costs this kernel scales with the bytes it moves (DESIGN_HISTORY.md 8.12).
//   * x edges: the flat neighbour of an image row's first / last pixel belongs to another row; those lanes' fragment addresses point at an all-zero row
//     of the halo slot instead (rows TP + 2 .. TP + 15 of the last piece are out-of-range lanes of the DMA, which writes zeros for them);
//   * y edges / ragged M: per-lane validity of the staged pixel per ky, taken from the tile pixel that reads it ("owner": halo row i belongs to pixel
//     clamp(i - 1)), SENT otherwise;
//   * LDS: A ring of four slots as before, halo ring of two (slot of group g = g & 1; it is refilled one K tile after its last read, the rule the A ring
//     follows), laid out [A0 A1 H0 | A2 A3 H1] so that the epilogue's staging region (from GROUP on) stays clear of the next tile's first two K tiles;
//   * halo rows are swizzled by ((row >> 2) & 1) << 1: conflict-free ds_read_b128 for 16 consecutive rows starting at ANY offset 0..2 of a 16-row block
//     (searched exhaustively over period-8 keys against the service groups quoted at swz64; the A operand's key is not: 2-way conflicts at shift 1, 2);
//   * every wave issues TP / 128 + 1 halo pieces with the kx = 0 tile of a group (the last one is real for wave 0 only, the others' all-SENT copy lands
//     in a dump KiB), so the counted vmcnt waits are compile-time constants per kx: the K loop is unrolled by three.
template <int TM, int TP, int WM, int WP, int NBUF, bool UPS, bool OUT_F32, bool KO, bool GEN = false, bool SUB = false, bool DYN = false, bool STATS = false,
          bool HALO = false>
__global__ __launch_bounds__(512) void conv_pp_kernel(Args a) {
#if __HIP_DEVICE_COMPILE__  // the host pass only needs the launch stub (hipcc drops the stub when it cannot digest the gfx950 body)
  constexpr int BM = TM / WM / 32, BP = TP / WP / 32;  // 32x32 accumulator blocks per wave
  constexpr int TILE_A = TM * 64, TILE_B = TP * 64, SLOT = TILE_A + TILE_B;
  constexpr int NPA = TM >= 128 ? TM / 128 : 1, NPB = TP / 128;  // 1-KiB DMA pieces per wave per K tile (TM = 64: the pieces of waves 4-7 are all padding and land in the dump KiB)
  static_assert(TM >= 128 || HALO, "the 64-row tile exists in the halo form only");
  constexpr int NP = NPA + NPB;
  constexpr int PF = NBUF - 1;  // prefetch distance in K tiles
  static_assert(WM * WP == 8 && BM * BP == 8, "8 waves x 8 accumulators");
  static_assert(!HALO || (!UPS && !GEN && !SUB && KO && NBUF == 4), "HALO: plain 3x3, chunk-outer K order");
  constexpr int NPH = NPB + 1;                     // halo pieces per wave and (chunk, ky) group
  constexpr int HALO_B = (TP + 16) * 64;           // halo slot: TP + 2 pixels, rows TP + 2 .. TP + 15 zero
  constexpr int GROUP = 2 * TILE_A + HALO_B;       // [A0 A1 H0] / [A2 A3 H1]
  constexpr int EPI_OFF = HALO ? GROUP : 2 * SLOT; // the epilogue's staging region
  constexpr int EPI_BYTES = 8 * 32 * (((BM >= 4 ? BM / 2 : BM) * 32) * 4 + 16);
  constexpr int DUMP_OFF = GROUP + (GROUP > EPI_BYTES ? GROUP : EPI_BYTES);
  // DIRECT (the HALO instantiations): the epilogue stores straight from the accumulators.  The MFMA runs with the PIXELS as its rows (D[pixel][cout]: lane
  // (c = lane & 15, g = lane >> 4) holds pixels 4g .. 4g + 3 of a 16-pixel block at cout column c of a 16-cout block) and the weight rows are DMA'd in a
  // permuted order (setup) so that a lane's BM16 accumulators of one pixel are BM16 consecutive couts and the 16 lanes of a group hold 16 BM16 consecutive
  // couts of that pixel: one store instruction writes 4 pixel rows x 256 (128) contiguous bytes with consecutive lanes on consecutive bytes -- no LDS
  // round trip (the staged epilogue wrote and re-read the tile in f32: 256 + 256 KB of LDS traffic and ~12 k cycles per 256 x 256 tile; first built and
  // measured in gemm_pp.hip, whose header has the numbers).  Same sums in the same order: results are bit-identical to the staged epilogue's.
  constexpr bool DIRECT = HALO && !OUT_F32;   // the HALO instantiations store straight from the accumulators (pixels as MFMA rows, permuted weight rows); the others stage through LDS
  constexpr int BM16 = BM * 2, BP16 = BP * 2;  // 16x16 MFMA blocks per wave (v_mfma_f32_16x16x32_bf16: measured 5 % less power per flop than 32x32x16,
                                               // tools/probes/probe_wavetile.hip arm D -- and the kernel is power-limited)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const int wm = wave / WP, wp = wave % WP;
  static_assert(!SUB || (!UPS && !GEN && KO), "SUB: its own instantiation, chunk-outer K order");
  const int T = SUB ? 4 : a.ks * a.ks;    // taps walked per output pixel
  const int TW = SUB ? 16 : T;            // taps per cout row of the packed weights
  const int nchunk = a.Cin >> 5;
  const int nK = T * nchunk;

  // ---- descriptors ---------------------------------------------------------------------------------------------
  const unsigned wbytes = (unsigned)a.Cout * TW * a.Cin * 2u;
  const unsigned xbytes = (unsigned)a.N * a.Hi * a.Wi * a.Cin * 2u;
  // Non-UPS gathers: a lane's base is its tap-0 source (by, bx) = (o * so - pd) for sd = 1, ceil((o - pd) / 2) for the zero-insertion gather
  // (sd = 2: tap k then reads by + (k >> 1), and only when o - pd + k is even), which can sit up to SR rows / columns outside the image;
  // the descriptor base is moved back by that much so that every lane offset and every wave-uniform tap offset is >= 0.
  const int SR = GEN ? ((UPS || a.ks == 1) ? 0 : (a.sd == 2 ? 1 : a.pd)) : 1;
  const int sds = GEN && a.sd == 2 ? 1 : 0;
  const unsigned shift = GEN ? ((!UPS && a.ks != 1) ? (unsigned)(SR * a.Wi + SR) * a.Cin * 2u : 0u)
                             : ((!UPS && (SUB || a.ks == 3)) ? (unsigned)(a.Wi + 1) * a.Cin * 2u : 0u);  // makes every tap offset >= 0
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, wbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB =
      __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(a.x) - shift), 0, xbytes + shift, 0x00020000);

  // Persistent blocks: one per CU, each walks tiles blockIdx.x, blockIdx.x + gridDim.x, ...  Re-dispatching a 512-thread
  // workgroup costs ~2 k cycles per XCD-serialised launch (tools/probes/time_conv_pp.py); gridDim.x is a multiple of 8, so a
  // block's tiles stay on one XCD's contiguous range of the XCD-aware order.  The per-tile address setup and the DMA of the
  // NEXT tile's first two K tiles are issued before the epilogue of the current tile, so the first-tile latency (3-8 k
  // cycles) hides under the store-bound epilogue (11-13 k cycles); the epilogue stages through ring slots 2.. for that.
  const int dvw = SUB ? a.Wi : a.Wo;   // pixel index -> (image, row, column) of the OUTPUT grid; SUB: of the source grid (one parity class of the output)
  const int hw = SUB ? a.Hi * a.Wi : a.Ho * a.Wo;
  const float inv_hw = 1.0f / (float)hw, inv_wo = 1.0f / (float)dvw;
  const bool small_m = a.M < (1 << 24);
  int m0 = 0, n0 = 0;  // tile whose DMA sources are currently set up
  int par = 0;         // SUB: output parity class (py << 1 | px) of that tile, wave-uniform
  unsigned voffA[NPA];
  unsigned ctrB[NPB], maskB[NPB], selB[NPB];
  unsigned rowo[UPS ? NPB : 1][3], colo[UPS ? NPB : 1][3];
  unsigned ctrH[HALO ? NPH : 1], mskH[HALO ? NPH : 1];   // HALO: byte offset of the lane's halo pixel (ky = 0 row), bit ky set when that row is inside the image
  int it_ky = 0;                                         // HALO: the issue state is (it_ch, it_ky) + the compile-time kx
  int it = 0, it_tap = 0, it_ch = 0;  // DMA issue state (wave-uniform): tile `it` = (tap it_tap, channel chunk it_ch)
  unsigned soffB_tap = 0;
  auto setup = [&](unsigned work) {
    int lane_s = lane;
    if constexpr (DIRECT) asm volatile("" : "+v"(lane_s));   // what setup derives from the lane index is recomputed per tile, not hoisted out of the tile loop and spilled
    const unsigned wid = xcd_remap(work, a.total);
    if constexpr (SUB) {  // order: source region, parity class, cout tile
      const unsigned q = wid / a.ctiles;
      n0 = (int)(wid - q * a.ctiles) * TM;
      par = __builtin_amdgcn_readfirstlane((int)(q & 3u));  // feeds the weight soffset: keep it on the scalar unit
      m0 = (int)(q >> 2) * TP;
    } else {
      m0 = (int)(wid / a.ctiles) * TP;  // first pixel
      n0 = (int)(wid % a.ctiles) * TM;  // first cout
    }
    it = it_tap = it_ch = 0;
    it_ky = 0;
#pragma unroll
    for (int p = 0; p < NPA; p++) {
      const int row = (wave * NPA + p) * 16 + (lane_s >> 2);
      int co = n0 + row;
      if constexpr (DIRECT) {   // LDS row 16 i + c of a wave's cout range holds cout BM16 * c + i: a lane's BM16 accumulators of one pixel are then consecutive couts
        constexpr int CW_ = TM / WM;
        const int wmr = row / CW_, rr = row % CW_;
        co = n0 + wmr * CW_ + BM16 * (rr & 15) + (rr >> 4);
      }
      const int c = (lane_s & 3) ^ swz64(row);  // logical 16-B chunk this lane fetches (LDS image stays lane-linear)
      voffA[p] = (co < a.Cout && row < TM) ? (unsigned)co * a.wsRow + c * 16u : SENT;     // wsRow / wsTap / wsChunk: tap-major [cout][T][cin] or K-tile-major [cin / 32][T][cout][32]
    }
    if constexpr (HALO) {
#pragma unroll
      for (int q = 0; q < NPH; q++) {
        const int i = (q < NPB ? (wave * NPB + q) * 16 : TP) + (lane_s >> 2);   // halo row = flat pixel m0 - 1 + i
        const int own = min(max(i - 1, 0), TP - 1);                            // the tile pixel whose edge rules it follows
        const int m = m0 + own;
        const int c = (lane_s & 3) ^ hswz(i);
        unsigned mask = 0;
        if (m < a.M && i < TP + 2 && (q < NPB || wave == 0)) {
          int n, r, y, x;
          divmod_small(m, hw, inv_hw, small_m, n, r);
          divmod_small(r, dvw, inv_wo, small_m, y, x);
          mask = (y > 0 ? 1u : 0u) | 2u | (y < a.Ho - 1 ? 4u : 0u);
          // the two outer halo pixels are only ever read from inside the owner's image row; where the flat neighbour belongs to another row it may lie
          // outside the tensor (the descriptor's range check does not see the scalar offset): not fetched
          if ((i == 0 && x == 0) || (i == TP + 1 && x == a.Wo - 1)) mask = 0;
        }
        ctrH[q] = (unsigned)(m0 + i) * a.Cin * 2u + c * 16u;   // relative to the descriptor base, which sits Wi + 1 pixels in front of the tensor
        mskH[q] = mask;
      }
    } else {
#pragma unroll
    for (int p = 0; p < NPB; p++) {
    const int row = (wave * NPB + p) * 16 + (lane_s >> 2);
    const int m = m0 + row;
    const int c = (lane_s & 3) ^ swz64(row);
    unsigned mask = 0;
    ctrB[p] = 0;
    if (m < a.M) {
      int n, r, y, x;
      divmod_small(m, hw, inv_hw, small_m, n, r);
      divmod_small(r, dvw, inv_wo, small_m, y, x);
      int by = y, bx = x;
      if constexpr (SUB) {  // bit a*2+b set when source pixel (by + a, bx + b) is inside the image
        by = y - 1 + (par >> 1); bx = x - 1 + (par & 1);
        const unsigned rm = (by >= 0 ? 0x3u : 0u) | (by + 1 < a.Hi ? 0xCu : 0u);
        const unsigned cm = (bx >= 0 ? 0x5u : 0u) | (bx + 1 < a.Wi ? 0xAu : 0u);
        mask = rm & cm;
      } else if (UPS || !GEN) {
        if (a.ks == 3) {  // bit ky*3+kx set when the tap stays inside the (upsampled) image
          const unsigned rm = (y > 0 ? 0x007u : 0u) | 0x038u | (y < a.Ho - 1 ? 0x1C0u : 0u);
          const unsigned cm = (x > 0 ? 0x049u : 0u) | 0x092u | (x < a.Wo - 1 ? 0x124u : 0u);
          mask = rm & cm;
        } else {
          mask = 1;
        }
      } else if (a.ks == 1) {
        mask = 1;
      } else {  // bit ky*ks+kx set when tap (ky, kx) reads a source pixel (inside the image, and on the grid for the zero-insertion gather)
        const int y0 = y * a.so - a.pd, x0 = x * a.so - a.pd;
        by = sds ? (y0 + 1) >> 1 : y0;
        bx = sds ? (x0 + 1) >> 1 : x0;
        unsigned rmk = 0, cmk = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          if (k < a.ks) {
            const bool ry = sds ? (((y0 + k) & 1) == 0 && (unsigned)(by + (k >> 1)) < (unsigned)a.Hi) : (unsigned)(by + k) < (unsigned)a.Hi;
            const bool rx = sds ? (((x0 + k) & 1) == 0 && (unsigned)(bx + (k >> 1)) < (unsigned)a.Wi) : (unsigned)(bx + k) < (unsigned)a.Wi;
            rmk |= (unsigned)ry << k;
            cmk |= (unsigned)rx << k;
          }
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
          if (k < a.ks && ((rmk >> k) & 1u)) mask |= cmk << (k * a.ks);
      }
      if (UPS) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const int yy = min(max(y + k - 1, 0), a.Ho - 1) >> 1, xx = min(max(x + k - 1, 0), a.Wo - 1) >> 1;
          rowo[UPS ? p : 0][k] = (unsigned)((n * a.Hi + yy) * a.Wi) * a.Cin * 2u + c * 16u;
          colo[UPS ? p : 0][k] = (unsigned)xx * a.Cin * 2u;
        }
        if (a.ks != 3) ctrB[p] = rowo[UPS ? p : 0][1] + colo[UPS ? p : 0][1];
      } else {
        if constexpr (GEN) ctrB[p] = (unsigned)((n * a.Hi + by + SR) * a.Wi + bx + SR) * a.Cin * 2u + c * 16u;
        else if constexpr (SUB) ctrB[p] = (unsigned)((n * a.Hi + by + 1) * a.Wi + bx + 1) * a.Cin * 2u + c * 16u;
        else ctrB[p] = (unsigned)((n * a.Hi + y) * a.Wi + x) * a.Cin * 2u + c * 16u;
      }
    }
    maskB[p] = mask;
    selB[p] = SENT;
  }
    }
  };

  // ---- fragment read offsets (bytes inside a slot) ------------------------------------------------------------------
  const int kg = lane >> 5;
  int aoff[BM16], boff[BP16];  // one 16-B read per lane covers a 16-row x 32-channel fragment (the whole K tile)
#pragma unroll
  for (int i = 0; i < BM16; i++) {
    const int row = wm * (TM / WM) + i * 16 + (lane & 15);
    aoff[i] = row * 64 + (((lane >> 4) ^ swz64(row)) << 4);
  }
#pragma unroll
  for (int j = 0; j < BP16; j++) {
    const int row = wp * (TP / WP) + j * 16 + (lane & 15);
    boff[j] = TILE_A + row * 64 + (((lane >> 4) ^ swz64(row)) << 4);
  }

  f32x4 acc[BM16][BP16];      // acc[i][j][r]: cout block i, row 4 * (lane >> 4) + r; pixel block j, column lane & 15

  auto new_tap = [&]() {
    int ky, kx;
    if constexpr (GEN) {
      ky = a.ks == 3 ? it_tap / 3 : (a.ks == 4 ? it_tap >> 2 : 1); kx = a.ks == 3 ? it_tap - (it_tap / 3) * 3 : (a.ks == 4 ? it_tap & 3 : 1);
      soffB_tap = (!UPS && a.ks != 1) ? (unsigned)((ky >> sds) * a.Wi + (kx >> sds)) * a.Cin * 2u : 0u;
    } else if constexpr (SUB) {
      ky = it_tap >> 1; kx = it_tap & 1;
      soffB_tap = (unsigned)(ky * a.Wi + kx) * a.Cin * 2u;
    } else {
      ky = a.ks == 3 ? it_tap / 3 : 1; kx = a.ks == 3 ? it_tap - (it_tap / 3) * 3 : 1;
      soffB_tap = (!UPS && a.ks == 3) ? (unsigned)(ky * a.Wi + kx) * a.Cin * 2u : 0u;
    }
#pragma unroll
    for (int p = 0; p < NPB; p++) {
      unsigned v = ctrB[p];
      if (UPS && a.ks == 3) {
        const unsigned ro = ky == 0 ? rowo[UPS ? p : 0][0] : (ky == 1 ? rowo[UPS ? p : 0][1] : rowo[UPS ? p : 0][2]);
        const unsigned co = kx == 0 ? colo[UPS ? p : 0][0] : (kx == 1 ? colo[UPS ? p : 0][1] : colo[UPS ? p : 0][2]);
        v = ro + co;
      }
      selB[p] = ((maskB[p] >> it_tap) & 1u) ? v : SENT;
    }
  };
  // issue this wave's pieces of tile `it` into the ring slot at byte offset `slot` (wave-uniform); past the last tile issue
  // all-zero pieces so that the vmcnt bookkeeping stays uniform (an out-of-range piece moves no memory)
  auto issue = [&](int slot) {
    const bool live = it < nK;
    if (live && (KO || it_ch == 0)) new_tap();
    unsigned soA = (unsigned)it_tap * a.wsTap + (unsigned)it_ch * a.wsChunk;
    if constexpr (SUB)  // tap (py + 2a, px + 2b) of the 4x4 operand
      soA = (unsigned)((((par >> 1) + (it_tap & 2)) << 2) + (par & 1) + ((it_tap & 1) << 1)) * a.wsTap + (unsigned)it_ch * a.wsChunk;
    unsigned soB = soffB_tap + (unsigned)it_ch * 64u;
    constexpr bool live_b = true;
    // GEN: the tap offset and the chunk counter end up in VGPRs (phis of VALU-computed values) and every piece issue became a readfirstlane
    // waterfall loop; pin the two wave-uniform offsets to SGPRs (the plain instantiation's code is unchanged)
    if constexpr (GEN) {
      soA = (unsigned)__builtin_amdgcn_readfirstlane((int)soA);
      soB = (unsigned)__builtin_amdgcn_readfirstlane((int)soB);
    }
#pragma unroll
    for (int p = 0; p < NPA; p++)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, LPTR(smem + slot + (wave * NPA + p) * 1024), 16, live ? voffA[p] : SENT, soA, 0, 0);
#pragma unroll
    for (int p = 0; p < NPB; p++)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, LPTR(smem + slot + TILE_A + (wave * NPB + p) * 1024), 16, (live && live_b) ? selB[p] : SENT, soB, 0, 0);
    it++;
    if (KO) {  // channel chunk outer, tap inner: the nine taps of a chunk re-read (shifted) the same activation lines back to back
      if (++it_tap == T) { it_tap = 0; it_ch++; }
    } else {
      if (++it_ch == nchunk) { it_ch = 0; it_tap++; }
    }
  };

  // HALO: K tile `it` = (chunk it_ch, ky it_ky, kx KX); its A pieces go to A slot it & 3, the group's halo pieces ride on the kx = 0 tile
  auto issue_h = [&](auto KXc) __attribute__((always_inline)) {
    constexpr int KX = decltype(KXc)::value;
    const bool live = it < nK;
    const unsigned soA = (unsigned)(it_ky * 3 + KX) * a.wsTap + (unsigned)it_ch * a.wsChunk;
    const int isl = it;
    const int da = ((isl >> 1) & 1) * GROUP + (isl & 1) * TILE_A;
#pragma unroll
    for (int p = 0; p < NPA; p++)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, LPTR(smem + ((wave * NPA + p) * 16 < TM ? da + (wave * NPA + p) * 1024 : DUMP_OFF)), 16, live ? voffA[p] : SENT, soA, 0, 0);
    if constexpr (KX == 0) {
      const unsigned soH = (unsigned)(it_ky * a.Wi) * a.Cin * 2u + (unsigned)it_ch * 64u;
      const int dh = ((it_ch + it_ky) & 1) * GROUP + 2 * TILE_A;
#pragma unroll
      for (int q = 0; q < NPH; q++) {
        const unsigned v = (live && ((mskH[HALO ? q : 0] >> it_ky) & 1u)) ? ctrH[HALO ? q : 0] : SENT;
        const int dst = q < NPB ? dh + (wave * NPB + q) * 1024 : (wave == 0 ? dh + TP * 64 : DUMP_OFF);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, LPTR(smem + dst), 16, v, soH, 0, 0);
      }
    }
    it++;
    if constexpr (KX == 2) {
      if (++it_ky == 3) { it_ky = 0; it_ch++; }
    }
  };
  using K0_ = std::integral_constant<int, 0>; using K1_ = std::integral_constant<int, 1>; using K2_ = std::integral_constant<int, 2>;

  // ---- persistent tile loop ---------------------------------------------------------------------------------------------
  auto stamp = [&](unsigned work, int k) {
    if (a.dbg && tid == 0) a.dbg[(size_t)work * 8 + k] = __builtin_amdgcn_s_memtime();
  };
  // DYN: thread 0 claims the tile AFTER the one being started and publishes it; every wave picks it up with the trailing DMA wait of the main loop
  auto claim = [&]() {
    if (tid == 0) {
      const unsigned x = blockIdx.x & 7u, first = gridDim.x >> 3;  // gridDim.x is a multiple of 8: `first` tiles per XCD range are the static first round
      unsigned flat = 0xFFFFFFFFu;
      for (unsigned k = 0; k < 8u; k++) {
        const unsigned xx = (x + k) & 7u;
        const unsigned cnt = ((unsigned)a.total >> 3) + (xx < ((unsigned)a.total & 7u) ? 1u : 0u);
        const unsigned j = first + atomicAdd(a.sched + xx, 1u);
        if (j < cnt) { flat = j * 8u + xx; break; }
      }
      __hip_atomic_store(a.sched + 16 + blockIdx.x, flat, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);  // read back by this block's own waves only
    }
  };
  stamp(blockIdx.x, 0);
  setup(blockIdx.x);
  stamp(blockIdx.x, 1);
  if constexpr (HALO) {
    issue_h(K0_{}); issue_h(K1_{}); issue_h(K2_{});
  } else {
#pragma unroll
    for (int u = 0; u < PF; u++) issue(u * SLOT);
  }
  for (unsigned work = blockIdx.x; work < (unsigned)a.total;) {
  const int m0c = m0, n0c = n0;  // the tile being computed (setup() moves m0 / n0 on to the next one before the epilogue)
  const int parc = par;
  if constexpr (DYN) claim();
  // HALO: fragment offsets of the three kx taps inside a halo slot -- pixel p reads halo row p + kx, or the slot's zero row where the tap leaves the image row
  int boffk[HALO ? 3 : 1][HALO ? BP16 : 1];
  if constexpr (HALO) {
    int lane_h = lane;
    asm volatile("" : "+v"(lane_h));   // computed here, per tile: not carried across the epilogue
#pragma unroll
    for (int j = 0; j < BP16; j++) {
      const int prow = wp * (TP / WP) + j * 16 + (lane_h & 15);
      int q, x;
      divmod_small(m0c + prow, dvw, inv_wo, small_m, q, x);
      const int zoff = (TP + 2) * 64 + ((lane_h >> 4) << 4);
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const int h = prow + k;
        const int off = h * 64 + (((lane_h >> 4) ^ hswz(h)) << 4);
        boffk[HALO ? k : 0][HALO ? j : 0] = (k == 0 && x == 0) || (k == 2 && x == a.Wo - 1) ? zoff : off;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < BM16; i++)
#pragma unroll
    for (int j = 0; j < BP16; j++)
#pragma unroll
      for (int r = 0; r < 4; r++) acc[i][j][r] = 0.f;
  wait_vmcnt<HALO ? 2 * NPA : (PF - 1) * NP>();
  __builtin_amdgcn_s_barrier();                // B_0: everybody's pieces of tile 0 have landed
  stamp(work, 2);
  if (grp == 1) __builtin_amdgcn_s_setprio(1);  // static priority for the second-dispatched half, no per-interval flips (MI355X_MICROARCH.md, two waves per SIMD, item 4)
  if (grp == 1) __builtin_amdgcn_s_barrier();  // stagger group 1 by one interval

  // ---- main loop ------------------------------------------------------------------------------------------------------
  bf16x8 af[BM16], bfr[BP16];
  int slot_rd = 0, slot_wr = PF * SLOT;
  unsigned next_dyn = 0;
  if constexpr (HALO) {
    int ka = 0, sh = 2 * TILE_A;   // A slot index and halo slot offset being read
    auto ktile = [&](auto KXc) __attribute__((always_inline)) {
      constexpr int KX = decltype(KXc)::value;
      const char* sa = smem + ((ka >> 1) & 1) * GROUP + (ka & 1) * TILE_A;
      const char* sb = smem + sh;
#pragma unroll
      for (int j = 0; j < BP16; j++) bfr[j] = *reinterpret_cast<const bf16x8*>(sb + boffk[HALO ? KX : 0][HALO ? j : 0]);
#pragma unroll
      for (int i = 0; i < BM16; i++) af[i] = *reinterpret_cast<const bf16x8*>(sa + aoff[i]);
      issue_h(KXc);   // tile t + 3: the same kx
      ka++;
      if constexpr (KX == 2) sh = sh == 2 * TILE_A ? GROUP + 2 * TILE_A : 2 * TILE_A;
      wait_vmcnt<KX == 2 ? 2 * NPA : 2 * NPA + NPH>();  // own pieces of the NEXT tile have landed: what may stay in flight is tiles t + 2 and t + 3
      __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0) through the builtin: the compiler sees the fragments have arrived and puts no lgkmcnt waits of its own between the MFMAs
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < BM16; i++)
#pragma unroll
        for (int j = 0; j < BP16; j++) {
          if constexpr (DIRECT) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(bfr[j]), "v"(af[i]));   // rows = pixels
          else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(af[i]), "v"(bfr[j]));
        }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    };
#pragma unroll 1
    for (int t = 0; t < nK; t += 3) { ktile(K0_{}); ktile(K1_{}); ktile(K2_{}); }
  } else {
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
    wait_vmcnt<(PF - 1) * NP>();  // own pieces of the NEXT tile have landed
    __builtin_amdgcn_s_waitcnt(0xC07F);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // COMPUTE interval (issuing the DMA from here, in the MFMA shadow, measured 5-8 % slower than from the LOAD interval)
#pragma unroll
    for (int i = 0; i < BM16; i++)
#pragma unroll
      for (int j = 0; j < BP16; j++)  // in-place accumulate in the AGPR half of the register file
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(af[i]), "v"(bfr[j]));
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  }
  stamp(work, 3);
  if (grp == 0) __builtin_amdgcn_s_barrier();  // matches group 1's extra barrier
  if constexpr (DYN) next_dyn = __hip_atomic_load(a.sched + 16 + blockIdx.x, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);  // published >= one barrier ago
  wait_vmcnt<0>();                             // the trailing all-zero pieces
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // MFMA (inline asm, invisible to the hazard recognizer) -> accumulator reads

  // ---- epilogue: accumulators -> LDS (f32, per-wave region) -> whole pixel rows, 16-B coalesced stores --------------------------
  // The MFMA layout gives a lane 4 couts of one pixel: stored directly, a wave-instruction scatters its lanes over 16 rows and the address path takes ~67
  // cycles for it whatever else the chip does (tools/probes/probe_store_bw.hip: 8.6 k cycles per 128 KB tile); 8 rows x 128 contiguous bytes with
  // consecutive lanes on consecutive bytes take 24 (3.0 k per tile from one CU, 4.8 k with all 256 CUs bursting).  So the tile is staged through LDS
  // and each store instruction writes RPI rows of CWH contiguous couts.  Phases per tile at 256->256 @128^2 (s_memtime stamps, tools/probes/time_conv_pp.py;
  // main loop 97.6 k cycles): ring free 0.7 k, bias + next tile's setup 2.2 k, its first two K tiles' DMA 2.0 k (queue time of the 64 KB), staging +
  // stores 6.4 k, drain 0.4 k; the staging uses ring slots 2.. in half-cout passes and leaves slots 0-1 to that DMA.
  __builtin_amdgcn_s_barrier();  // every wave's trailing DMA has landed and all fragment reads are done: the ring is free
  stamp(work, 6);
  const unsigned next = DYN ? (unsigned)__builtin_amdgcn_readfirstlane((int)next_dyn) : work + gridDim.x;
  const bool has_next = next < (unsigned)a.total;
  if constexpr (DIRECT) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    constexpr int CL = BM16;               // consecutive couts per lane: 8 (256-row tile) or 4
    constexpr int NQD = CL / 4;            // 4-channel quads per lane (GroupNorm statistics)
    static_assert(CL == 8 || CL == 4, "DIRECT: 8 or 4 couts per lane");
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));       // recomputed per tile, not carried (spilled) across the main loop
    const int cc = lane_o & 15, gq = lane_o >> 4;
    const int col = n0c + wm * (TM / WM) + CL * cc;
    const bool c_ok = col < a.Cout;        // Cout % 8 == 0: a lane's couts are all inside or all outside
    const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (unsigned)a.M * (unsigned)a.Cout * 2u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rBias = __builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0, a.bias ? (unsigned)a.Cout * 4u : 0u, 0x00020000);
    // bias first, while the memory queue is empty (behind the next tile's prefetch its first use would wait for those pieces to land)
    float bsv[CL];
    {
      const unsigned vo = c_ok ? (unsigned)col * 4u : SENT;
#pragma unroll
      for (int h = 0; h < NQD; h++) {
        const u32x4 b4 = __builtin_amdgcn_raw_buffer_load_b128(rBias, vo == SENT ? SENT : vo + 16u * h, 0, 0);
        const f32x4 bf = *reinterpret_cast<const f32x4*>(&b4);
#pragma unroll
        for (int e = 0; e < 4; e++) bsv[4 * h + e] = bf[e];
      }
    }
    if (has_next) {
      stamp(next, 0);
      setup(next);
      stamp(next, 1);
    } else {
      it = (nK + 3) & ~3;   // destination derived from `it`: A slot 0 / halo slot 0
      it_ky = it_ch = 0;
    }
    issue_h(K0_{});
    stamp(work, 7);
    float s1[STATS ? NQD : 1], s2[STATS ? NQD : 1];
    if constexpr (STATS) {
#pragma unroll
      for (int h = 0; h < NQD; h++) { s1[h] = 0.f; s2[h] = 0.f; }
    }
    // the lane's first row (pixel block 0, r = 0) as a byte offset -- out of range when its couts are -- and the wave-uniform step of one pixel row
    const unsigned vbase = c_ok ? ((unsigned)(m0c + wp * (TP / WP) + 4 * gq) * (unsigned)a.Cout + (unsigned)col) * 2u : SENT;
    const unsigned rstep = (unsigned)a.Cout * 2u;
    auto body = [&](auto RESc, auto ACTc) __attribute__((always_inline)) {
      constexpr bool RES = decltype(RESc)::value;
      constexpr int ACTC = decltype(ACTc)::value;   // < 0: a.act is read at run time (the activations that are not hot)
      const __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc((void*)a.res, 0, RES ? (unsigned)a.M * (unsigned)a.Cout * 2u : 0u, 0x00020000);
      // residual / gate operand: the four rows of pixel block j + 1 are fetched while block j is converted and stored
      typedef std::conditional_t<CL == 8, u32x4, u32x2> rvec;
      rvec rnx[RES ? 4 : 1];   // a rolling window of four rows: row r of block j + 1 is requested as soon as row r of block j has been consumed
      auto fetch = [&](int j, int r) {
        if constexpr (RES) {
          if constexpr (CL == 8) rnx[RES ? r : 0] = __builtin_amdgcn_raw_buffer_load_b128(rR, vbase, (unsigned)(j * 16 + r) * rstep, 2);
          else rnx[RES ? r : 0] = __builtin_amdgcn_raw_buffer_load_b64(rR, vbase, (unsigned)(j * 16 + r) * rstep, 2);
        }
      };
#pragma unroll
      for (int r = 0; r < 4; r++) fetch(0, r);
#pragma unroll
      for (int j = 0; j < BP16; j++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          float v[CL];
#pragma unroll
          for (int i = 0; i < CL; i++) {
            // read where it is used: left to itself the compiler hoists the accumulator reads every variant of `body` shares in front of the variant
            // dispatch -- dozens of values live across the whole epilogue, spilled to scratch, and every scratch reload is a VMEM load whose wait drains the
            // next tile's prefetch queue
            float x;
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(acc[i][j][r]));
            v[i] = x + bsv[i];
          }
          const int act = ACTC >= 0 ? ACTC : a.act;
          if constexpr (RES) {
            float rf[CL];
            {
              const rvec rv = rnx[RES ? r : 0];
#pragma unroll
              for (int h = 0; h < CL / 2; h++) {
                const unsigned wd = rv[h];
                const bf16x2 b2 = *reinterpret_cast<const bf16x2*>(&wd);
                rf[2 * h] = (float)b2[0]; rf[2 * h + 1] = (float)b2[1];
              }
            }
            if (j + 1 < BP16) fetch(j + 1, r);
            if (act == 3) {   // ReLU-backward gate: `res` is the saved activation, not an addend
#pragma unroll
              for (int e = 0; e < CL; e++) v[e] = rf[e] > 0.f ? v[e] : 0.f;
            } else {
#pragma unroll
              for (int e = 0; e < CL; e++) v[e] += rf[e];
            }
          }
          if (act == 1) {
#pragma unroll
            for (int e = 0; e < CL; e++) v[e] = v[e] * sigmoidf_(v[e]);
          } else if (act == 2) {
#pragma unroll
            for (int e = 0; e < CL; e++) v[e] = v[e] > 0.f ? v[e] : 0.f;
          } else if (act == 4) {
#pragma unroll
            for (int e = 0; e < CL; e++) v[e] = v[e] > 0.f ? v[e] : 0.2f * v[e];
          }
          const unsigned so = (unsigned)(j * 16 + r) * rstep;   // wave-uniform row step; rows past M fall off the descriptor's end (tests/test_gpu_conv_bounds.py)
          unsigned pk[CL / 2];
#pragma unroll
          for (int h = 0; h < CL / 2; h++) pk[h] = dmvae_pack_bf16x2(v[2 * h], v[2 * h + 1]);
          if constexpr (STATS) {   // v_dot2c_f32_bf16 on the packed result: rows past M do not occur (whole pixel tiles per image), couts past Cout carry exact zeros
            const bf16x2 one2 = {(bf16)1.0f, (bf16)1.0f};
#pragma unroll
            for (int h = 0; h < CL / 2; h++) {
              const bf16x2 p2 = *reinterpret_cast<const bf16x2*>(&pk[h]);
              s1[STATS ? h >> 1 : 0] = __builtin_amdgcn_fdot2_f32_bf16(p2, one2, s1[STATS ? h >> 1 : 0], false);
              s2[STATS ? h >> 1 : 0] = __builtin_amdgcn_fdot2_f32_bf16(p2, p2, s2[STATS ? h >> 1 : 0], false);
            }
          }
          // non-temporal: the tile is next read by a later kernel, after far more than an L2 of other traffic
          if constexpr (CL == 8) {
            const u32x4 o = {pk[0], pk[1], pk[2], pk[3]};
            __builtin_amdgcn_raw_buffer_store_b128(o, rY, vbase, so, 2 /* nt */);
            asm volatile("s_nop 0" :: "v"(o));   // gfx950: a VALU write to a store's data VGPR directly behind the store is seen by the store (see the staged epilogue)
          } else {
            const u32x2 o = {pk[0], pk[1]};
            __builtin_amdgcn_raw_buffer_store_b64(o, rY, vbase, so, 2 /* nt */);
            asm volatile("s_nop 0" :: "v"(o));
          }
        }
      }
    };
    using T_ = std::true_type; using F_ = std::false_type;
    if (a.res) {
      if (a.act == 0) body(T_{}, std::integral_constant<int, 0>{});
      else if (a.act == 3) body(T_{}, std::integral_constant<int, 3>{});
      else body(T_{}, std::integral_constant<int, -1>{});
    } else {
      if (a.act == 0) body(F_{}, std::integral_constant<int, 0>{});
      else if (a.act == 2) body(F_{}, std::integral_constant<int, 2>{});
      else body(F_{}, std::integral_constant<int, -1>{});
    }
    if constexpr (STATS) {   // the four lane groups hold different pixel rows of the same couts: fold them, then one partial per (pixel tile, wave column, quad)
      const size_t trow = (size_t)(m0c / TP) * WP + wp;
#pragma unroll
      for (int h = 0; h < NQD; h++) {
        float t1 = s1[STATS ? h : 0], t2 = s2[STATS ? h : 0];
        t1 += __shfl_xor(t1, 16, 64); t2 += __shfl_xor(t2, 16, 64);
        t1 += __shfl_xor(t1, 32, 64); t2 += __shfl_xor(t2, 32, 64);
        const int cq = col + 4 * h;
        if (gq == 0 && cq < a.Cout) *reinterpret_cast<f32x2*>(a.gnpart + (trow * (a.Cout >> 2) + (cq >> 2)) * 2) = f32x2{t1, t2};
      }
    }
  } else {
    // 64 couts (two 32-cout blocks) per staging pass: a store instruction then writes 8 pixel rows x 128 contiguous bytes.  The 128-row tile (one wave =
    // 64 couts) used to take them in two passes of 32 -- 16 rows x 64 B per store instruction, i.e. twice the cache lines per instruction -- and now takes
    // them in one.
    //
    // Straight-line code, no exec-masked branches: rows past M / couts past Cout are out-of-range offsets of buffer descriptors (stores dropped, loads
    // return 0), the residual operand and the activation are compile-time cases.  The first form of this epilogue tested `m < M && c_ok` around each
    // store and loaded the residual under `if (a.res)`; with VMEM operations inside conditional blocks the compiler's wait-count pass could not count
    // them and put `s_waitcnt vmcnt(0)` behind every pair of staging reads -- every one of the 16 stores of a wave waited for the previous store's
    // acknowledgement from memory (17-19 k cycles per 128 KB tile, 12-24 % of the kernel; the stores themselves need ~3 k at the rate
    // tools/probes/probe_store_bw.hip measures with every CU bursting).  A wave's LDS operations execute in order, so the staging region needs no
    // waits of its own either: the next round's accumulators are written right behind this round's reads.
    constexpr int EH = BM >= 4 ? BM / 2 : BM;   // cout blocks per staging pass
    constexpr int NH = BM / EH;              // staging passes over the wave's couts
    constexpr int CWH = EH * 32;             // couts per wave per pass
    constexpr int ROWB = CWH * 4 + 16;       // padded f32 row (bank-conflict-free ds_write_b128)
    constexpr int LPR = CWH / 8, RPI = 64 / LPR;
    constexpr int NI = 32 / RPI;             // store instructions per round (32 pixel rows)
    constexpr unsigned ES = OUT_F32 ? 4u : 2u;
    static_assert(!STATS || !OUT_F32, "STATS: statistics of the bf16 result");
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    char* reg = smem + EPI_OFF + wave * (32 * ROWB);
    // an opaque copy of the lane index: everything the epilogue derives from it is recomputed per tile instead of being hoisted out of the tile loop and
    // carried (spilled) across the main loop
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));
    const int cl = lane_o % LPR, rg = lane_o / LPR;
    const unsigned mout = SUB ? 4u * (unsigned)a.M : (unsigned)a.M;   // output pixels (the host keeps mout * Cout * ES below 2^31: SENT stays out of range)
    const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, mout * (unsigned)a.Cout * ES, 0x00020000);
    const __amdgpu_buffer_rsrc_t rBias = __builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0, a.bias ? (unsigned)a.Cout * 4u : 0u, 0x00020000);
    // bias of every pass, loaded while the memory queue is empty: behind the next tile's prefetch (vmcnt counts in order) the first use below would wait
    // for those LDS-DMA pieces to land -- 2-3 k cycles per tile
    f32x4 bias_lo[NH], bias_hi[NH];
#pragma unroll
    for (int hh = 0; hh < NH; hh++) {
      const int cb = n0c + wm * (TM / WM) + hh * CWH + cl * 8;
      const unsigned vo = cb < a.Cout ? (unsigned)cb * 4u : SENT;
      const u32x4 b0 = __builtin_amdgcn_raw_buffer_load_b128(rBias, vo, 0, 0), b1 = __builtin_amdgcn_raw_buffer_load_b128(rBias, vo + 16u, 0, 0);
      bias_lo[hh] = *reinterpret_cast<const f32x4*>(&b0);
      bias_hi[hh] = *reinterpret_cast<const f32x4*>(&b1);
    }
    // The next tile's setup and its first K tile go out before the stores, its second K tile behind them (all eight waves' pieces of two K tiles in front of
    // the stores held every wave at the issue for 2-3 k cycles; the second tile is not read before the main loop's second iteration).  The first issue is
    // unconditional (past the last tile: all-zero pieces), so that the compiler can count the operations between the bias loads and their first use
    // instead of draining the queue at a control-flow join.
    if (has_next) {
      stamp(next, 0);
      setup(next);
      stamp(next, 1);
    } else {
      it = HALO ? (nK + 3) & ~3 : nK;   // HALO derives the destination from `it`: A slot 0 / halo slot 0, clear of the staging region
      it_ky = it_ch = 0;
    }
    if constexpr (HALO) issue_h(K0_{}); else issue(0);
    stamp(work, 7);
    float sacc[STATS ? NH : 1][2][2];  // [pass][4-channel half of the lane's 8 couts][sum, sum of squares]
    if constexpr (STATS) {
#pragma unroll
      for (int i = 0; i < NH * 4; i++) (&sacc[0][0][0])[i] = 0.f;
    }
    auto body = [&](auto RESc, auto ACTc) __attribute__((always_inline)) {
      constexpr bool RES = decltype(RESc)::value;
      constexpr int ACTC = decltype(ACTc)::value;   // < 0: a.act is read at run time (the activations that are not hot)
      const __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc((void*)a.res, 0, RES ? mout * (unsigned)a.Cout * 2u : 0u, 0x00020000);
#pragma unroll
      for (int hh = 0; hh < NH; hh++) {
        if (n0c + wm * (TM / WM) + hh * CWH >= a.Cout) continue;  // this wave's couts of the pass are all padding (wave-uniform; e.g. Cout = 64 on the 128-row tile)
        const int cb = n0c + wm * (TM / WM) + hh * CWH + cl * 8;  // this lane's 8 couts in the read phase
        const bool c_ok = cb < a.Cout;
        // Addresses.  Plain tiles: the lane's row (rg, couts cb..) has ONE byte offset per pass -- SENT when its couts are past Cout -- and every later row of
        // the tile is a wave-uniform multiple of the row stride further (the instruction's scalar offset; rows past M fall off the descriptor's end).
        // SUB: output pixels are parity-interleaved, so each row's offset is computed (two divmods).
        const unsigned row_e = (unsigned)(m0c + wp * (TP / WP) + rg) * (unsigned)a.Cout + (unsigned)cb;
        const unsigned ybase = c_ok ? row_e * ES : SENT, rbase = c_ok ? row_e * 2u : SENT;
        const unsigned ystep = (unsigned)(RPI * a.Cout) * ES, rstep = (unsigned)(RPI * a.Cout) * 2u;   // wave-uniform
        auto sub_off = [&](int j, int it2) -> unsigned {   // element offset of the lane's row for the per-parity kernel, or SENT
          const int m = m0c + wp * (TP / WP) + j * 32 + it2 * RPI + rg;
          int n, r, y, x;
          divmod_small(m, hw, inv_hw, small_m, n, r);
          divmod_small(r, dvw, inv_wo, small_m, y, x);
          const unsigned mo = (unsigned)((n * a.Ho + 2 * y + (parc >> 1)) * a.Wo + 2 * x + (parc & 1));
          return (m < a.M && c_ok) ? mo * (unsigned)a.Cout + (unsigned)cb : SENT;
        };
        auto stage = [&](int j) {   // accumulators of pixel block j -> the wave's staging region
#pragma unroll
          for (int i = 0; i < EH * 2; i++)       // 16-cout blocks of this pass
#pragma unroll
            for (int jb = 0; jb < 2; jb++)       // the two 16-pixel blocks of pixel block j
              *reinterpret_cast<f32x4*>(reg + (jb * 16 + (lane_o & 15)) * ROWB + (i * 16 + 4 * (lane_o >> 4)) * 4) = acc[hh * EH * 2 + i][j * 2 + jb];
        };
        // Register budget (128 VGPRs beside the 128 accumulators, two waves per SIMD): with a residual operand the staging reads are taken in two halves
        // of NI / 2 rows and the operand is held one round ahead (its loads go out before the previous round's stores: vmcnt counts in order, a load queued
        // behind a store waits for that store's acknowledgement).  Holding whole rounds of reads, offsets and residuals spilled 3-58 VGPRs to scratch.
        constexpr int NPART = RES ? 2 : 1, NQ = NI / NPART;
        u32x4 r8s[RES ? 2 : 1][RES ? NI : 1];
        auto fetch = [&](int j, int h2) {   // residual / gate operand of rows h2 * NQ .. of round j
          if constexpr (RES) {
#pragma unroll
            for (int it2 = h2 * NQ; it2 < (h2 + 1) * NQ; it2++) {
              if constexpr (SUB) {
                const unsigned eo = sub_off(j, it2);
                r8s[j & 1][it2] = __builtin_amdgcn_raw_buffer_load_b128(rR, eo == SENT ? SENT : eo * 2u, 0, 2);
              } else {
                r8s[j & 1][it2] = __builtin_amdgcn_raw_buffer_load_b128(rR, rbase, (unsigned)(j * NI + it2) * rstep, 2);
              }
            }
          }
        };
#pragma unroll
        for (int h2 = 0; h2 < NPART; h2++) fetch(0, h2);
        stage(0);
#pragma unroll
        for (int j = 0; j < BP; j++) {
#pragma unroll
          for (int h2 = 0; h2 < NPART; h2++) {
          f32x4 lo[NQ], hi[NQ];
#pragma unroll
          for (int q = 0; q < NQ; q++) {
            const int px = (h2 * NQ + q) * RPI + rg;
            lo[q] = *reinterpret_cast<const f32x4*>(reg + px * ROWB + cl * 32);
            hi[q] = *reinterpret_cast<const f32x4*>(reg + px * ROWB + cl * 32 + 16);
          }
          if (j + 1 < BP) {
            if (h2 == NPART - 1) stage(j + 1);   // in-order LDS: these writes land behind the reads above (all of round j's reads have been issued)
            fetch(j + 1, h2);
          }
          u32x4 keep[OUT_F32 ? 2 * NQ : NQ];
#pragma unroll
          for (int q = 0; q < NQ; q++) {
            const int it2 = h2 * NQ + q;
            f32x4 v0 = lo[q] + bias_lo[hh], v1 = hi[q] + bias_hi[hh];
            float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            const int act = ACTC >= 0 ? ACTC : a.act;
            if constexpr (RES) {
              const bf16x8 r8 = *reinterpret_cast<const bf16x8*>(&r8s[j & 1][it2]);
              if (act == 3) {  // ReLU-backward gate: `res` is the saved activation, not an addend
#pragma unroll
                for (int e = 0; e < 8; e++) v[e] = (float)r8[e] > 0.f ? v[e] : 0.f;
              } else {
#pragma unroll
                for (int e = 0; e < 8; e++) v[e] += (float)r8[e];
              }
            }
            if (act == 1) {
#pragma unroll
              for (int e = 0; e < 8; e++) v[e] = v[e] * sigmoidf_(v[e]);
            } else if (act == 2) {
#pragma unroll
              for (int e = 0; e < 8; e++) v[e] = v[e] > 0.f ? v[e] : 0.f;
            } else if (act == 4) {
#pragma unroll
              for (int e = 0; e < 8; e++) v[e] = v[e] > 0.f ? v[e] : 0.2f * v[e];
            }
            unsigned voff = ybase, soff = (unsigned)(j * NI + it2) * ystep;
            if constexpr (SUB) {
              const unsigned eo = sub_off(j, it2);
              voff = eo == SENT ? SENT : eo * ES;
              soff = 0u;
            }
            if constexpr (OUT_F32) {
              const f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
              keep[2 * q] = *reinterpret_cast<const u32x4*>(&o0);
              keep[2 * q + 1] = *reinterpret_cast<const u32x4*>(&o1);
              __builtin_amdgcn_raw_buffer_store_b128(keep[2 * q], rY, voff, soff, 0);
              __builtin_amdgcn_raw_buffer_store_b128(keep[2 * q + 1], rY, voff + 16u, soff, 0);
            } else {
              const u32x4 o = {dmvae_pack_bf16x2(v[0], v[1]), dmvae_pack_bf16x2(v[2], v[3]), dmvae_pack_bf16x2(v[4], v[5]), dmvae_pack_bf16x2(v[6], v[7])};
              if constexpr (STATS) {  // v_dot2c_f32_bf16 on the packed result: two channels per instruction, 8 instructions per store instead of 24
                // rows past M do not occur (whole pixel tiles per image), couts past Cout carry exact zeros (zero weights, zero bias, zero residual)
                const bf16x2 one2 = {(bf16)1.0f, (bf16)1.0f};
#pragma unroll
                for (int pr = 0; pr < 4; pr++) {
                  const unsigned pw = o[pr];
                  const bf16x2 p2 = *reinterpret_cast<const bf16x2*>(&pw);
                  float& s1 = sacc[STATS ? hh : 0][pr >> 1][0];
                  float& s2 = sacc[STATS ? hh : 0][pr >> 1][1];
                  s1 = __builtin_amdgcn_fdot2_f32_bf16(p2, one2, s1, false);
                  s2 = __builtin_amdgcn_fdot2_f32_bf16(p2, p2, s2, false);
                }
              }
              // non-temporal: the tile is next read by a later kernel, after far more than an L2 of other traffic
              keep[q] = o;
              __builtin_amdgcn_raw_buffer_store_b128(o, rY, voff, soff, 2);
            }
          }
          // gfx950 hazard that hipcc (ROCm 7.2) does not pad: a VALU instruction that directly follows a buffer_store_dwordx4 and writes one of its data
          // VGPRs is seen by the store in lanes 12-15 of every 16 -- also when the scalar offset is an SGPR, the case LLVM's hazard recogniser exempts
          // (tools/probes/probe_store_war.hip: 0.2-5 % of the dwords clobbered at distance 0, none with one wait state).  With 8 VALU instructions per
          // row the next row's v_pk_add_f32 landed right behind the store and its sums in the previous row's output (deterministic per build, found by
          // tools/probes/dbg_epi2.py).  Every row's packed result of the part stays live up to the s_nop below, so no store's data can be written before it.
#pragma unroll
          for (int q = 0; q < (OUT_F32 ? 2 * NQ : NQ); q++) asm volatile("s_nop 0" :: "v"(keep[q]));
          }
        }
      }
    };
    using T_ = std::true_type; using F_ = std::false_type;
    if (a.res) {
      if (a.act == 0) body(T_{}, std::integral_constant<int, 0>{});
      else if (a.act == 3) body(T_{}, std::integral_constant<int, 3>{});
      else body(T_{}, std::integral_constant<int, -1>{});
    } else {
      if (a.act == 0) body(F_{}, std::integral_constant<int, 0>{});
      else if (a.act == 2) body(F_{}, std::integral_constant<int, 2>{});
      else body(F_{}, std::integral_constant<int, -1>{});
    }
    if constexpr (STATS) {  // lanes that share `cl` hold different pixel rows of the same 8 couts: fold them, then one partial per (tile, wave column, quad)
      const size_t trow = (size_t)(SUB ? (m0c / TP) * 4 + parc : m0c / TP) * WP + wp;
#pragma unroll
      for (int hh = 0; hh < NH; hh++)
#pragma unroll
        for (int qd = 0; qd < 2; qd++) {
          float s1 = sacc[STATS ? hh : 0][qd][0], s2 = sacc[STATS ? hh : 0][qd][1];
#pragma unroll
          for (int off = LPR; off < 64; off <<= 1) { s1 += __shfl_xor(s1, off, 64); s2 += __shfl_xor(s2, off, 64); }
          const int cq = n0c + wm * (TM / WM) + hh * CWH + lane_o * 8 + qd * 4;
          if (lane_o < LPR && cq < a.Cout) *reinterpret_cast<f32x2*>(a.gnpart + (trow * (a.Cout >> 2) + (cq >> 2)) * 2) = f32x2{s1, s2};
        }
    }
  }
  if (has_next) { if constexpr (HALO) issue_h(K1_{}); else issue(SLOT); }
  stamp(work, 4);
  wait_vmcnt<HALO ? NPA : NP>();              // the epilogue's stores share vmcnt with the prefetched K tiles: everything but the second tile's pieces (the newest) has landed
  stamp(work, 5);
  __builtin_amdgcn_s_barrier();  // staging reads done before ring slots 2.. are refilled
  if (has_next) { if constexpr (HALO) issue_h(K2_{}); else issue(2 * SLOT); }
  work = next;
  }  // persistent tile loop
  if constexpr (DYN) {
    if (tid == 0 && atomicAdd(a.sched + 8, 1u) == gridDim.x - 1) {  // last block out: nobody claims any more -- leave the words zero for the next launch
#pragma unroll
      for (int k = 0; k < 9; k++) __hip_atomic_store(a.sched + k, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
#endif
}

// Scheduling words of the DYN instantiation: one set per stream (launches on one stream never overlap; sets are self-cleaning, so a captured graph replays
// correctly).  Static device memory of the code object, zero at load: the library allocates nothing.
constexpr int SCHED_WORDS = 16 + 512, SCHED_SLOTS = 32;
__device__ unsigned g_sched[SCHED_SLOTS * SCHED_WORDS];

// Keyed by (device, stream): a __device__ symbol has one address PER DEVICE, and a process that drives several GPUs must not hand one device's kernel the
// scheduling words of another.
static unsigned* sched_for(hipStream_t st) {
  static std::mutex mu;
  static std::unordered_map<unsigned long long, int> slot;     // (device, stream) -> set index on that device
  static std::unordered_map<int, unsigned*> base;              // device -> address of g_sched there
  static std::unordered_map<int, int> used;                    // device -> sets handed out
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  std::lock_guard<std::mutex> lk(mu);
  auto b = base.find(dev);
  if (b == base.end()) {
    unsigned* p = nullptr;
    if (hipGetSymbolAddress(reinterpret_cast<void**>(&p), HIP_SYMBOL(g_sched)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    b = base.emplace(dev, p).first;
  }
  const unsigned long long key = ((unsigned long long)(unsigned)dev << 56) ^ (unsigned long long)reinterpret_cast<uintptr_t>(st);
  auto it = slot.find(key);
  if (it == slot.end()) {
    int& n = used[dev];
    if (n >= SCHED_SLOTS) return nullptr;  // more streams than sets on this device: those launches keep the static stride
    it = slot.emplace(key, n++).first;
  }
  return b->second + (size_t)it->second * SCHED_WORDS;
}

// Dynamic tile claiming on / off for every instantiation at once: -1 = not set (DMVAE_PP_DYNAMIC decides, read once), 0 / 1 = dmvae_set_dynamic.
static std::atomic<int> g_dynamic{-1};
static bool dynamic_on() {
  const int v = g_dynamic.load(std::memory_order_relaxed);
  if (v >= 0) return v != 0;
  static const bool env = [] { const char* e = getenv("DMVAE_PP_DYNAMIC"); return e ? atoi(e) != 0 : false; }();
  return env;
}

template <int TM, int TP, int WM, int WP, int NBUF, bool UPS, bool F32, bool KO, bool GEN = false, bool SUB = false, bool DYN = false, bool STATS = false,
          bool HALO = false>
int launch(Args a, hipStream_t st) {
  a.ctiles = (a.Cout + TM - 1) / TM;
  a.total = ((a.M + TP - 1) / TP) * a.ctiles * (SUB ? 4 : 1);
  constexpr int persist = 256;      // persistent blocks: one per CU
  const unsigned grid = (persist > 0 && a.total > persist) ? (unsigned)persist : (unsigned)a.total;
  if constexpr (!DYN && !UPS && KO && !F32) {  // the bf16-output, chunk-outer instantiations (every large launch of the training step) have a DYN twin
    if (dynamic_on() && grid == 256u && (unsigned)a.total > grid) {
      a.sched = sched_for(st);
      if (a.sched) return launch<TM, TP, WM, WP, NBUF, UPS, F32, KO, GEN, SUB, true, STATS, HALO>(a, st);
    }
  }
  constexpr int cwh = (TM / WM / 32 >= 4) ? TM / WM / 2 : TM / WM;    // couts per staging pass (the kernel's CWH)
  constexpr int ring = NBUF * (TM + TP) * 64, epi = 2 * (TM + TP) * 64 + 8 * 32 * (cwh * 4 + 16);
  constexpr int group = 2 * TM * 64 + (TP + 16) * 64, epi_h = 8 * 32 * (cwh * 4 + 16);   // the kernel's GROUP / EPI_BYTES / DUMP_OFF
  constexpr int lds = HALO ? group + (group > epi_h ? group : epi_h) + 1024 : (ring > epi ? ring : epi);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_pp_kernel<TM, TP, WM, WP, NBUF, UPS, F32, KO, GEN, SUB, DYN, STATS, HALO>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done = true;
  }
  hipLaunchKernelGGL((conv_pp_kernel<TM, TP, WM, WP, NBUF, UPS, F32, KO, GEN, SUB, DYN, STATS, HALO>), dim3(grid), dim3(512), lds, st, a);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

// KO (last template argument) = K-tile order: false = tap outer / channel chunk inner, true = chunk outer / tap inner.  With the taps
// inner the nine shifted re-reads of a 32-channel chunk follow each other within ~10 us and hit in the XCD's L2; with the taps outer each
// tap streams the whole channel depth of the pixel tile (256 KB per CU at Cin = 512) and the next tap finds nothing of it left:
// FETCH_SIZE per launch 9.0x vs 2.2x the compulsory bytes at 512->512 @128^2 (profiles/r1_conv_hbm_traffic.txt).  The folded-upsample
// variant keeps the taps outer (its per-tap source selection is too costly to redo every K tile).
static constexpr int halo_mode() { return 3; }   // bit 0: the 256 x 256 tile, bit 1: + the 128 x 512 and 64 x 1024 tiles run the kx-halo form
static constexpr bool korder_on() { return true; }   // channel chunk outer, tap inner (DESIGN_HISTORY.md 3.1: the taps-outer order re-fetched the input 9x from HBM)
// the launches pick() sends to a HALO instantiation (given that dmvae_conv_pp_try takes the shape at all)
static bool halo_for(int ks, int cout, bool plain, bool ups, bool f32) {
  const int h = halo_mode();
  return plain && !ups && !f32 && ks == 3 && h != 0 && korder_on() && (cout > 128 || (h & 2));
}

template <bool UPS, bool F32>
int pick(const Args& a, hipStream_t st, bool gen) {
  if constexpr (!UPS) {
    if (gen) {  // strided / 4x4 / zero-insertion gathers: chunk-outer K order only
      if (a.Cout <= 128) return launch<128, 512, 2, 4, 4, false, F32, true, true>(a, st);
      return launch<256, 256, 2, 4, 4, false, F32, true, true>(a, st);
    }
  }
  if constexpr (UPS) {
    if (a.Cout <= 128) return launch<128, 512, 2, 4, 4, UPS, F32, false>(a, st);
    return launch<256, 256, 2, 4, 4, UPS, F32, false>(a, st);
  } else {
    if constexpr (!F32) {
      if (halo_for(a.ks, a.Cout, !gen, UPS, F32)) {   // plain 3x3: the three kx taps of a (chunk, ky) share one staged halo of the pixel tile
        if (a.Cout > 128) return a.gnpart ? launch<256, 256, 2, 4, 4, false, false, true, false, false, false, true, true>(a, st)
                                          : launch<256, 256, 2, 4, 4, false, false, true, false, false, false, false, true>(a, st);
        if (a.Cout <= 64 && !a.gnpart)   // LPIPS trunk's 64-channel layers (lpips.py:116-153): a 128-row tile would be half padding
          return launch<64, 1024, 1, 8, 4, false, false, true, false, false, false, false, true>(a, st);
        return a.gnpart ? launch<128, 512, 2, 4, 4, false, false, true, false, false, false, true, true>(a, st)
                        : launch<128, 512, 2, 4, 4, false, false, true, false, false, false, false, true>(a, st);
      }
      if (a.gnpart) {  // GroupNorm statistics of the result in the epilogue (chunk-outer K order only)
        if (a.Cout <= 128) return launch<128, 512, 2, 4, 4, false, false, true, false, false, false, true>(a, st);
        return launch<256, 256, 2, 4, 4, false, false, true, false, false, false, true>(a, st);
      }
    }
    if (a.Cout <= 128) return launch<128, 512, 2, 4, 4, UPS, F32, true>(a, st);  // 160 KiB of LDS: the whole CU
    return launch<256, 256, 2, 4, 4, UPS, F32, true>(a, st);
  }
}

}  // namespace dmvae_conv_pp

// Entry used by dmvae_conv2d_nhwc_fwd (conv_fwd.hip) for the shapes this kernel covers; returns 1 when it declines.
static unsigned long long* g_dbg = nullptr;
extern "C" void dmvae_debug_timing(void* buf) { g_dbg = (unsigned long long*)buf; }  // diagnostics only (tools/probes/time_conv_pp.py)

// Diagnostics only (tools/probes/contention.py): `blocks` workgroups that each hold `lds_bytes` of LDS and spin for `microseconds` -- a stand-in for another
// stream's resident kernel (an RCCL collective) taking CUs away from the persistent conv blocks.
__global__ __launch_bounds__(64) void occupy_kernel(unsigned long long ticks) {
  extern __shared__ char occ[];
  occ[threadIdx.x] = 0;
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
// Dynamic tile claiming (DYN) for every conv_pp launch from now on: 1 = on, 0 = off, -1 = back to the DMVAE_PP_DYNAMIC environment default.  One process-wide
// switch (the first version latched the environment variable per template instantiation at its first launch: instantiations that had run before
// dmvae_amd.dist set the variable stayed static).  Returns the value in force (0 / 1).
extern "C" int dmvae_set_dynamic(int on) {
  dmvae_conv_pp::g_dynamic.store(on < 0 ? -1 : (on != 0), std::memory_order_relaxed);
  return dmvae_conv_pp::dynamic_on() ? 1 : 0;
}

extern "C" int dmvae_debug_occupy(int blocks, int lds_bytes, int microseconds, hipStream_t stream) {
  static bool attr_done = false;
  if (!attr_done) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(occupy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_done = true; }
  hipLaunchKernelGGL(occupy_kernel, dim3(blocks), dim3(64), lds_bytes, stream, (unsigned long long)microseconds * 100ull);  // wall_clock64: 100 MHz
  DMVAE_CHECK_LAUNCH();
  return 0;
}

int dmvae_conv_geometry(const dmvae_conv_desc* d, int* ho, int* wo, int* so, int* pd, int* sd, int* fl);  // conv_fwd.hip

// The gates of dmvae_conv_pp_try below + halo_for(): 1 when the descriptor runs on a HALO instantiation (and may therefore carry w_layout = 1).
extern "C" int dmvae_conv_halo_applies(const dmvae_conv_desc* d) {
  using namespace dmvae_conv_pp;
  if (!d) return 0;
  int ho, wo, so, pd, sd, fl;
  if (dmvae_conv_geometry(d, &ho, &wo, &so, &pd, &sd, &fl) != 0) return 0;
  const bool plain = !(d->upsample == 2 || d->stride == 2 || d->ks == 4 || d->transposed);
  const long long M = (long long)d->n * ho * wo;
  const long long xbytes = (long long)d->n * d->h * d->w * d->cin * 2;
  const long long wbytes = (long long)d->cout * d->ks * d->ks * d->cin * 2;
  constexpr long long min_m = 16384;      // smaller problems: conv_fwd.hip
  if (d->cin % 32 != 0 || d->cout < 64 || d->cout % 8 != 0 || M < min_m || xbytes + (1ll << 22) >= (1ll << 31) || wbytes >= (1ll << 31)) return 0;
  if (M >= (1ll << 24) || M * d->cout * (d->out_f32 ? 4 : 2) >= (1ll << 31)) return 0;
  return halo_for(d->ks, d->cout, plain, fl != 0, d->out_f32 != 0) ? 1 : 0;
}

// 1 when the descriptor runs on this file's kernel at all (any instantiation): it may then carry w_layout = 1 -- the per-parity / general-gather / 4x4 instantiations
// read the K-tile-major operand through the same three strides as the halo ones.  (ks = 1 operands have no K-tile-major copy on the host side.)
extern "C" int dmvae_conv_kmajor_applies(const dmvae_conv_desc* d) {
  using namespace dmvae_conv_pp;
  if (!d) return 0;
  int ho, wo, so, pd, sd, fl;
  if (dmvae_conv_geometry(d, &ho, &wo, &so, &pd, &sd, &fl) != 0) return 0;
  const bool plain = !(d->upsample == 2 || d->stride == 2 || d->ks == 4 || d->transposed);
  const long long M = (long long)d->n * ho * wo;
  const long long xbytes = (long long)d->n * d->h * d->w * d->cin * 2;
  const long long wbytes = (long long)d->cout * d->ks * d->ks * d->cin * 2;
  constexpr long long min_m = 16384;      // smaller problems: conv_fwd.hip
  if (d->cin % 32 != 0 || d->cout < 64 || d->cout % 8 != 0 || M < min_m || xbytes + (1ll << 22) >= (1ll << 31) || wbytes >= (1ll << 31)) return 0;
  if (M >= (1ll << 24) || M * d->cout * (d->out_f32 ? 4 : 2) >= (1ll << 31)) return 0;
  return 1;
}

// gnpart / gn_groups: when non-null and the shape allows (bf16 result, plain or per-parity route, whole pixel tiles per image, groups of a multiple of 4
// channels) the launch also leaves per-tile GroupNorm partials there and *gn_tp is set to the pixel-tile size (else 0: the caller computes the statistics itself).
int dmvae_conv_pp_try(const void* x, const void* w, const void* bias, const void* residual, void* y, const dmvae_conv_desc* d,
                      hipStream_t stream, float* gnpart, int gn_groups, int* gn_tp) {
  if (gn_tp) *gn_tp = 0;
  using namespace dmvae_conv_pp;
  int ho, wo, so, pd, sd, fl;
  if (dmvae_conv_geometry(d, &ho, &wo, &so, &pd, &sd, &fl) != 0) return 1;
  const bool plain = !(d->upsample == 2 || d->stride == 2 || d->ks == 4 || d->transposed);
  const bool sub = d->transposed && d->ks == 4 && d->stride == 2;
  const int ups = fl ? 1 : 0;      // nearest x2 folded into the gather: its own template variant
  const long long M = (long long)d->n * ho * wo;
  const long long xbytes = (long long)d->n * d->h * d->w * d->cin * 2;
  const long long wbytes = (long long)d->cout * d->ks * d->ks * d->cin * 2;
  constexpr long long min_m = 16384;      // smaller problems: conv_fwd.hip
  if (d->cin % 32 != 0 || d->cout < 64 || d->cout % 8 != 0 || M < min_m || xbytes + (1ll << 22) >= (1ll << 31) || wbytes >= (1ll << 31)) return 1;
  if (M >= (1ll << 24)) return 1;   // divmod_small
  if (M * d->cout * (d->out_f32 ? 4 : 2) >= (1ll << 31)) return 1;   // the epilogue addresses y (and the residual) through 32-bit buffer offsets; SENT must stay out of range
  if (d->w_layout != 0 && d->w_layout != 1) {
    dmvae_set_error("conv2d_nhwc_fwd: w_layout %d (0 = tap-major, 1 = K-tile-major where dmvae_conv_kmajor_applies(d) is 1)", d->w_layout);
    return -1;
  }
  Args a;
  a.wsRow = d->w_layout ? 64u : (unsigned)(d->ks * d->ks * d->cin) * 2u;
  a.wsTap = d->w_layout ? (unsigned)d->cout * 64u : (unsigned)d->cin * 2u;
  a.wsChunk = d->w_layout ? (unsigned)(d->ks * d->ks) * (unsigned)d->cout * 64u : 64u;
  a.x = (const bf16*)x; a.w = (const bf16*)w; a.bias = (const float*)bias; a.res = (const bf16*)residual; a.y = y;
  a.N = d->n; a.Hi = d->h; a.Wi = d->w; a.Cin = d->cin; a.Cout = d->cout;
  a.Ho = ho; a.Wo = wo; a.so = so; a.pd = pd; a.sd = sd;
  a.ks = d->ks; a.act = d->act; a.M = (int)M; a.ctiles = 0; a.dbg = g_dbg;
  a.Ml = d->n * d->h * d->w;
  a.gnpart = nullptr;
  if (gnpart && gn_tp && !d->out_f32 && (plain || sub) && !ups && gn_groups > 0 && d->cout % gn_groups == 0 && (d->cout / gn_groups) % 4 == 0) {
    const int tp = d->cout <= 128 ? 512 : 256;
    const long long per_image = sub ? (long long)d->h * d->w : (long long)ho * wo;   // pixels of one image per pixel-tile sequence (per parity class for SUB)
    if (per_image % tp == 0) { a.gnpart = gnpart; *gn_tp = tp; }
  }
  a.sched = nullptr;
  const bool f32 = d->out_f32 != 0;
  if (sub) {  // per-parity 2x2 decomposition: pixel tiles run over the source grid, once per parity class
    a.M = a.Ml;
    if (a.gnpart) {
      if (a.Cout <= 128) return launch<128, 512, 2, 4, 4, false, false, true, false, true, false, true>(a, stream);
      return launch<256, 256, 2, 4, 4, false, false, true, false, true, false, true>(a, stream);
    }
    if (a.Cout <= 128) return f32 ? launch<128, 512, 2, 4, 4, false, true, true, false, true>(a, stream) : launch<128, 512, 2, 4, 4, false, false, true, false, true>(a, stream);
    return f32 ? launch<256, 256, 2, 4, 4, false, true, true, false, true>(a, stream) : launch<256, 256, 2, 4, 4, false, false, true, false, true>(a, stream);
  }
  if (ups) return f32 ? pick<true, true>(a, stream, false) : pick<true, false>(a, stream, false);
  return f32 ? pick<false, true>(a, stream, !plain) : pick<false, false>(a, stream, !plain);
}
