// Weight gradient of a 3x3 stride-1 convolution with at most FOUR output channels on gfx950 -- the decoder's conv_out (models/flux_ae.py:237,274:
// 128 -> 3 channels at the image resolution):
//
//   dW[co][ci][ky][kx] = sum_q a[q][ci] * dy[q - (ky - 1, kx - 1)][co]                 (reduction over 2 M pixels, 3456 results)
//
// A 128-row matrix tile is 98 % padding here, and the route this replaces (im2col of the gradient to [pixels][72] + the general TN GEMM) moved 840 MB through a
// kernel that ran at 75 TFLOP/s (495 + 75 us).  The contraction is shaped by its bytes instead: `a` is read ONCE (537 MB at [32, 256^2, 128]); the gradient is
// 3 channels x 2 B per pixel and is prepared (wgrad_thin_pack_kernel) as three x-shifted planar copies with a zero row above and below,
//   sh[kx][n][co][1 + y][x] = dy[n][co][y][x - kx + 1]   (bf16, zeros outside the image),
// so that every tap's operand row is an aligned, unmasked 64-B segment.
//
//   GEMM view per block (a range of 32-pixel K steps):  G[ci 0..127][col = tap * 4 + co, 36 of 48] += A^T B,  v_mfma_f32_16x16x32_bf16
//   - A operand = a^T: the [32 px][128 ch] tile is staged by LDS-DMA exactly as conv_wgrad_pp.hip stages its operands (32-B slots XOR-swizzled by the pixel row)
//     and read with ds_read_b64_tr_b16; wave w owns channel blocks 2 w, 2 w + 1.
//   - B operand = the 36 (tap, co) segments of 32 pixels, 2304 B per K step, three LDS-DMA pieces; lane (col, pixel octet) reads its 16 B directly.
//   - one s_barrier per K step, a 6-deep ring (66 KiB: two workgroups per CU), counted vmcnt -- the queue is never drained.
//   Per-block partial sums go to a slab [block][48][128]; wgrad_thin_reduce_kernel adds them in fixed order into PyTorch's [co][ci][ky][kx] layout.
// HBM-bound: 256 B of `a` + 24 B of gradient copies per pixel.
#include "common.h"
#include "dmvae_hip.h"
#include <cstdlib>

namespace dmvae_wgrad_thin {

constexpr unsigned SENT = 0x80000000u;
constexpr int CIN = 128, NCOL = 48, NBUF = 6, PF = 5;
constexpr int SLOT_A = 32 * 256, SLOT_B = 3 * 1024, SLOT = SLOT_A + SLOT_B;

struct Args {
  const bf16* a;    // [N, H, W, 128]
  const bf16* sh;   // [3][N][4][H + 2][W]
  float* slab;      // [blocks][48][128]
  int N, H, W;
  int steps, spb;   // K steps in all (N * H * W / 32), per block
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  __builtin_amdgcn_s_waitcnt((N & 15) | 0x0F70 | ((N >> 4) << 14));
  asm volatile("" ::: "memory");
}
// inline asm, not the builtin: no memory operand for the compiler's wait-count pass to order behind the LDS-DMA queue (see conv_wgrad_pp.hip)
template <int OFF>
__device__ __forceinline__ s16x4 tr_read(unsigned lds_addr) {
  s16x4 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(lds_addr), "n"(OFF));
  return r;
}
__device__ __forceinline__ bf16x8 lds_read16(unsigned lds_addr) {
  bf16x8 r;
  asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(lds_addr));
  return r;
}

// dy [N][C][H][W] f32 (C <= 4 real channels) -> sh [3][N][4][H + 2][W] bf16
// One thread per eight consecutive x of one (n, co, row): the gradient row is read ONCE (two aligned 16-B loads + the two neighbours) and all three x-shifted
// copies are written from it as 16-B stores.  (First form: one element of one copy per thread -- four runtime integer divisions and a 4-B load per 2-B store,
// the gradient read three times: 52 us for 25 M elements.)  W % 32 == 0 (host).
__global__ __launch_bounds__(256) void wgrad_thin_pack_kernel(const float* __restrict__ dy, bf16* __restrict__ sh, int N, int C, int H, int W) {
  const unsigned w8 = (unsigned)W >> 3;
  const unsigned per = (unsigned)N * 4u * (H + 2) * w8;     // 16-B pieces of one copy; 3 per < 2^31 / 8 (host check)
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < per; i += gridDim.x * 256u) {
    const unsigned x0 = (i % w8) * 8u;
    unsigned r = i / w8;
    const unsigned yy = r % (unsigned)(H + 2); r /= (unsigned)(H + 2);
    const unsigned co = r & 3u, n = r >> 2;
    const int y = (int)yy - 1;
    float v[10];      // v[1 + e] = dy[x0 + e], v[0] / v[9]: the neighbours (zero outside the row)
#pragma unroll
    for (int e = 0; e < 10; e++) v[e] = 0.f;
    if ((int)co < C && (unsigned)y < (unsigned)H) {
      const float* row = dy + (((size_t)n * C + co) * H + y) * W + x0;
      const f32x4 lo = *reinterpret_cast<const f32x4*>(row), hi = *reinterpret_cast<const f32x4*>(row + 4);
#pragma unroll
      for (int e = 0; e < 4; e++) { v[1 + e] = lo[e]; v[5 + e] = hi[e]; }
      if (x0 > 0) v[0] = row[-1];
      if (x0 + 8 < (unsigned)W) v[9] = row[8];
    }
#pragma unroll
    for (int kx = 0; kx < 3; kx++) {      // sh[kx][..][x] = dy[x - kx + 1]
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; e++) o[e] = (bf16)v[1 + e - kx + 1];
      reinterpret_cast<bf16x8*>(sh)[(size_t)kx * per + i] = o;
    }
  }
}

__global__ __launch_bounds__(256) void wgrad_thin_kernel(Args a) {
#if __HIP_DEVICE_COMPILE__
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int s0 = blockIdx.x * a.spb, s1 = min(s0 + a.spb, a.steps);
  const int nK = s1 - s0;
  const int xpr = a.W >> 5;  // K steps per image row

  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)a.a, 0, (unsigned)a.N * a.H * a.W * 256u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)a.sh, 0, 3u * (unsigned)a.N * 4u * (unsigned)(a.H + 2) * (unsigned)a.W * 2u, 0x00020000);

  // ---- per-lane DMA sources ------------------------------------------------------------------------------------------------------------
  // A: pieces pb = 2 wave, 2 wave + 1 of the [32 px][128 ch] tile: pixel rows 4 pb + lane / 16, physical 16-B chunk lane % 16 (swizzle of conv_wgrad_pp.hip)
  unsigned voffA[2];
#pragma unroll
  for (int p = 0; p < 2; p++) {
    const int pb = wave * 2 + p, row = pb * 4 + (lane >> 4), cphys = lane & 15;
    const int clog = ((((cphys >> 1) ^ (((row & 3) << 1) | ((row >> 3) & 1))) << 1) | (cphys & 1)) * 8;
    voffA[p] = (unsigned)(row * CIN + clog) * 2u;
  }
  // B: piece `wave` (waves 0..2) of the 36 segments: segment s = 16 piece + lane / 4 = tap * 4 + co, 16-B chunk lane % 4 (8 pixels)
  unsigned voffB = SENT;
  {
    const int seg = wave * 16 + (lane >> 2), ch = lane & 3;
    if (wave < 3 && seg < 36) {
      const int tap = seg >> 2, co = seg & 3, ky = tap / 3, kx = tap - ky * 3;
      voffB = (unsigned)((((kx * a.N) * 4 + co) * (a.H + 2) + (2 - ky)) * a.W) * 2u + (unsigned)ch * 16u;
    }
  }

  // ---- fragment read addresses --------------------------------------------------------------------------------------------------------------------
  const int G = lane >> 4, rr = (lane & 15) >> 2, qq = lane & 3;
  const int fkey = (rr << 1) | (G & 1);
  unsigned aoff[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int c = (wave * 2 + i) * 16 + 4 * qq;
    aoff[i] = (unsigned)((G * 8 + rr) * 256 + ((((c >> 4) & 7) ^ fkey) << 5) + (c & 15) * 2);
  }
  unsigned boff[3];
#pragma unroll
  for (int j = 0; j < 3; j++) boff[j] = (unsigned)(SLOT_A + (j * 16 + (lane & 15)) * 64 + G * 16);

  f32x4 acc[2][3];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
      for (int r = 0; r < 4; r++) acc[i][j][r] = 0.f;

  // ---- DMA issue state: K step `st` = (n, y, x block), wave-uniform --------------------------------------------------------------------------------
  int it = 0, st = s0;
  int xb, py, pn;
  {  // division results pinned to SGPRs: everything derived from them (the DMA's wave-uniform offsets) then stays on the scalar unit
    const int rows = __builtin_amdgcn_readfirstlane(s0 / xpr);
    xb = s0 - rows * xpr;
    pn = __builtin_amdgcn_readfirstlane(rows / a.H);
    py = rows - pn * a.H;
  }
  auto issue = [&](int slot) {
    const bool live = it < nK;
    const unsigned soA = (unsigned)st * (32u * 256u);                                  // 32 consecutive pixels of one image row
#pragma unroll
    for (int p = 0; p < 2; p++)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, LPTR(smem + slot + (wave * 2 + p) * 1024), 16, live ? voffA[p] : SENT, soA, 0, 0);
    const unsigned soB = (unsigned)__builtin_amdgcn_readfirstlane((((pn * 4) * (a.H + 2) + py) * a.W + xb * 32) * 2);   // wave-uniform: keep it off the VALU (no waterfall loop)
    // wave 3 has no B piece: its third issue is fully masked and lands in a scratch KiB behind the ring (same count per wave for the vmcnt protocol)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, LPTR(smem + (wave < 3 ? slot + SLOT_A + wave * 1024 : NBUF * SLOT)), 16, live ? voffB : SENT, soB, 0, 0);
    it++; st++;
    if (++xb == xpr) { xb = 0; if (++py == a.H) { py = 0; pn++; } }
  };
#pragma unroll
  for (int u = 0; u < PF; u++) issue(u * SLOT);
  const unsigned lds0 = (unsigned)(size_t)LPTR(smem);
  int slot_rd = 0, slot_wr = PF * SLOT;
#pragma unroll 1
  for (int t = 0; t < nK; t++) {
    wait_vmcnt<(PF - 1) * 3>();        // this wave's pieces of step t have landed
    __builtin_amdgcn_s_barrier();      // ... and everybody else's; every wave is also done reading step t - 1's slot
    const unsigned sb = lds0 + (unsigned)slot_rd;
    union { bf16x8 v; s16x4 h[2]; } af[2];
    bf16x8 bfr[3];
#pragma unroll
    for (int i = 0; i < 2; i++) {
      af[i].h[0] = tr_read<0>(sb + aoff[i]);
      af[i].h[1] = tr_read<1024>(sb + aoff[i]);
    }
#pragma unroll
    for (int j = 0; j < 3; j++) bfr[j] = lds_read16(sb + boff[j]);
    issue(slot_wr);                    // into step t - 1's slot
    slot_rd = slot_rd + SLOT == NBUF * SLOT ? 0 : slot_rd + SLOT;
    slot_wr = slot_wr + SLOT == NBUF * SLOT ? 0 : slot_wr + SLOT;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 3; j++)
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(af[i].v), "v"(bfr[j]));
  }
  wait_vmcnt<0>();
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // MFMA (inline asm) -> accumulator reads

  // ---- slab [block][col][ci]: lane owns column (l & 15) of block j, 4 channels per accumulator ---------------------------------------------------------
  float* slab = a.slab + (size_t)blockIdx.x * NCOL * CIN;
#pragma unroll
  for (int j = 0; j < 3; j++) {
    const int col = j * 16 + (lane & 15);
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int ci = (wave * 2 + i) * 16 + 4 * G;
      *reinterpret_cast<f32x4*>(slab + (size_t)col * CIN + ci) = acc[i][j];
    }
  }
#endif
}

// dW[co][ci][ky][kx] (+)= sum_blocks slab[b][(ky * 3 + kx) * 4 + co][ci]; one block per (tap, co) row, 8 groups of 128 channel threads share the blocks
__global__ __launch_bounds__(1024) void wgrad_thin_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int blocks, int cout, int accumulate) {
  __shared__ float red[8][CIN];
  const int col = blockIdx.x, tap = col >> 2, co = col & 3;
  const int ci = threadIdx.x & 127, g = threadIdx.x >> 7;
  const int per = (blocks + 7) / 8, b0 = g * per, b1 = min(b0 + per, blocks);
  float s = 0.f;
  for (int b = b0; b < b1; b++) s += slab[((size_t)b * NCOL + col) * CIN + ci];
  red[g][ci] = s;
  __syncthreads();
  if (g == 0 && co < cout) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; k++) t += red[k][ci];
    float* o = dw + ((size_t)co * CIN + ci) * 9 + tap;
    *o = (accumulate ? *o : 0.f) + t;
  }
}

}  // namespace dmvae_wgrad_thin

// Workspace: the three shifted gradient copies + the slab.  0 when the shape is not this kernel's.
static int wgrad_thin_blocks(long long steps) {
  int spb = (int)((steps + 511) / 512);
  if (spb < 8) spb = 8;
  return (int)((steps + spb - 1) / spb);
}
extern "C" size_t dmvae_conv_out_wgrad_workspace(int n, int h, int w, int cin, int cout) {
  if (cin != 128 || cout < 1 || cout > 4 || w % 32 != 0 || n < 1 || h < 1 || (long long)n * h * w * 256 >= (1ll << 31) ||
      3ll * n * 4 * (h + 2) * w * 2 >= (1ll << 31))
    return 0;
  const long long steps = (long long)n * h * w / 32;
  return (size_t)3 * n * 4 * (h + 2) * w * 2 + 256 + (size_t)wgrad_thin_blocks(steps) * dmvae_wgrad_thin::NCOL * dmvae_wgrad_thin::CIN * sizeof(float);
}

// dy: [N][cout][H][W] f32 (the image gradient as autograd hands it over), a: [N][H][W][128] bf16 (the conv's input), dw: [cout][128][3][3] f32.
extern "C" int dmvae_conv_out_wgrad(const void* dy, const void* a, void* dw, void* workspace, size_t workspace_bytes, int n, int h, int w, int cin, int cout,
                                    int accumulate, hipStream_t stream) {
  using namespace dmvae_wgrad_thin;
  DMVAE_CHECK_ARG(dy && a && dw && workspace, "conv_out_wgrad: null pointer");
  const size_t need = dmvae_conv_out_wgrad_workspace(n, h, w, cin, cout);
  DMVAE_CHECK_ARG(need > 0, "conv_out_wgrad: unsupported shape n=%d h=%d w=%d cin=%d cout=%d (Cin 128, Cout <= 4, W %% 32 == 0)", n, h, w, cin, cout);
  DMVAE_CHECK_ARG(workspace_bytes >= need, "conv_out_wgrad: workspace too small (%zu < %zu)", workspace_bytes, need);
  bf16* sh = (bf16*)workspace;
  const size_t sh_bytes = ((size_t)3 * n * 4 * (h + 2) * w * 2 + 255) & ~(size_t)255;
  float* slab = (float*)((char*)workspace + sh_bytes);
  hipLaunchKernelGGL(wgrad_thin_pack_kernel, dim3(2048), dim3(256), 0, stream, (const float*)dy, sh, n, cout, h, w);
  DMVAE_CHECK_LAUNCH();
  Args g;
  g.a = (const bf16*)a; g.sh = sh; g.slab = slab; g.N = n; g.H = h; g.W = w;
  g.steps = (int)((long long)n * h * w / 32);
  const int blocks = wgrad_thin_blocks(g.steps);
  g.spb = (g.steps + blocks - 1) / blocks;
  constexpr int lds = NBUF * SLOT + 1024;   // + wave 3's scratch KiB
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_thin_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done = true;
  }
  hipLaunchKernelGGL(wgrad_thin_kernel, dim3((unsigned)blocks), dim3(256), lds, stream, g);
  DMVAE_CHECK_LAUNCH();
  hipLaunchKernelGGL(wgrad_thin_reduce_kernel, dim3(36), dim3(1024), 0, stream, (const float*)slab, (float*)dw, blocks, cout, accumulate);
  DMVAE_CHECK_LAUNCH();
  return 0;
}
