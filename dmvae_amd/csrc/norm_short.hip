// A ResnetBlock's 1x1 `nin_shortcut` (models/flux_ae.py:67,77-82) riding on the block's first GroupNorm passes.
//
// In a ResnetBlock whose channel count changes (the first block of up[1]: 512 -> 256 at 128^2, and of up[0]: 256 -> 128 at 256^2), the shortcut is a 1x1 conv of
// the block input x -- the tensor norm1 reads in the forward -- and its input gradient dy Ws is added to the gradient norm1's backward writes.  As launches of
// the conv kernel both are HBM-bound on tensors the GroupNorm passes stream anyway (forward: x read a second time, 1.6 GB in 332 us at 256 -> 128 @ 256^2;
// backward: the 1074-MB gradient written by the conv and read back by the GroupNorm backward's apply pass as `dres`, 340 + ~90 us).  Here the passes evaluate the
// 1x1 conv in place on the matrix cores, from the 16-pixel x 256-channel tiles they hold:
//   forward  (short_apply_kernel):     a = swish(GN(x)) -> bf16,  xs = bf16(x Ws^T + bs)          reads x once, writes a and xs
//   backward (short_bwd_apply_kernel): dx = GN'(da, x) + bf16(dy Ws)                              reads da, x and dy (half the channels of the stored gradient)
// with the stored tensor's bf16 rounding kept where it was (xs is what conv2's epilogue adds; the shortcut gradient is rounded before it is added), so either
// result differs from the stored-operand route only through the order in which one element's products are added in f32.
//
// A wave owns 16-pixel runs.  The weights (64 KB of bf16 at 256 <-> 128 channels) sit in LDS as the A fragments of v_mfma_f32_16x16x32_bf16, [fragment][lane]
// (row = lane & 15 is an output channel, k = 8 (lane >> 4) .. + 7 a reduction channel), filled once per block; the run's pixels are the B operand (column =
// lane & 15 is a pixel), read from a per-wave LDS tile that the wave fills with 1-KB contiguous global loads.  D (lane <-> pixel p, channels 16 j + 4 kg + i)
// goes back through the tile and returns in the elementwise layout -- lane <-> pixel 2 r + (lane >> 5), channels 8 (lane & 31) + 0..7 -- in which x / da are
// loaded and a / dx stored as 1-KB contiguous wave accesses (csrc/groupnorm.hip::convout_bwd_kernel: the MFMA's own layout as the access pattern cost more than
// the arithmetic).  One block of 8 waves per CU (133 KB of LDS): the memory stream is kept full by issuing a run's 16 loads before its MFMA phase and the next
// run's small operand before that; per run 64 MFMAs and 64 KB of fragment reads from LDS against 20-24 KB of HBM traffic.
//
// 512 <-> 256 channels (up[1]): the weight is 256 KB -- more than a CU's LDS.  There the block's eight waves own it in REGISTERS (128 accumulation-file registers
// per lane: a wave holds the A fragments of its slice of the output channels; the MFMA reads them in place) and work on ONE 16-pixel run together: the run's
// pixels go through a shared LDS tile as the B operand of every wave, each wave's slice of D comes back through a second tile, two workgroup barriers per run;
// the next run's global loads are issued a whole run ahead (coop_apply_kernel / coop_bwd_apply_kernel).
#include "common.h"
#include "dmvae_hip.h"

namespace dmvae_ns {

constexpr int CB = 256, CS = 128;             // channels of the block input (x, da, dx, a) and of the shortcut's other side (xs, dy)
constexpr int NW = 8;                         // waves per block
constexpr int W_BYTES = CB * CS * 2;          // the 1x1 weight as bf16
constexpr int BROW = CB * 2 + 16;             // bytes per pixel row of a 16 x CB tile (b64 / b128 accesses of 16 rows: conflict-free)
constexpr int SROW = CS * 2 + 16;             // ... of a 16 x CS tile; both tiles of a run share the wave's 16 * BROW bytes (never live together)
constexpr int TILE = 16 * BROW;
constexpr int LDS_BYTES = W_BYTES + NW * TILE;   // 133120

struct Geom {
  int HW, G, cpg, ppc, nchunk;
};

__device__ __forceinline__ bf16x8 ldnt(const bf16* p) { return __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p)); }
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// A fragments of W [M][K] (row-major bf16, K contiguous) into LDS: fragment (j, s) = rows 16 j .. + 15, k = 32 s .. + 31; KS = K / 32 fragments per row block
template <int M, int K>
__device__ __forceinline__ void fill_weights(char* wl, const bf16* __restrict__ w) {
  constexpr int KS = K / 32;
  for (int i = threadIdx.x; i < (M / 16) * KS * 64; i += NW * 64) {
    const int f = i >> 6, ll = i & 63, j = f / KS, s = f - j * KS;
    *reinterpret_cast<bf16x8*>(wl + i * 16) = *reinterpret_cast<const bf16x8*>(w + (size_t)(16 * j + (ll & 15)) * K + 32 * s + 8 * (ll >> 4));
  }
}

// ---- forward: a = act(GN(x)), xs = x Ws^T + bs ----------------------------------------------------------------------------------------------------------
template <int ACT>
__global__ __launch_bounds__(NW * 64) void short_apply_kernel(const bf16* __restrict__ x, const float* __restrict__ stats, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const bf16* __restrict__ w, const float* __restrict__ bias,
                                                             bf16* __restrict__ a, bf16* __restrict__ xs, Geom g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  fill_weights<CS, CB>(smem, w);
  const int wv = threadIdx.x >> 6, l = threadIdx.x & 63, p = l & 15, kg = l >> 4;
  const int lc = l & 31, hp = l >> 5;                       // elementwise layout: channels 8 lc .. + 7 of pixel 2 r + hp
  const int n = blockIdx.y;
  char* tile = smem + W_BYTES + wv * TILE;
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; e++) {
    const int c = 8 * lc + e;
    const float* st = stats + ((size_t)n * g.G + c / g.cpg) * 2;
    sc[e] = st[1] * gamma[c];
    sh[e] = beta[c] - st[0] * sc[e];
  }
  f32x4 bs[CS / 16];
#pragma unroll
  for (int j = 0; j < CS / 16; j++)
#pragma unroll
    for (int i = 0; i < 4; i++) bs[j][i] = bias ? bias[16 * j + 4 * kg + i] : 0.f;
  __syncthreads();
  const int p0 = blockIdx.x * g.ppc, p1 = min(p0 + g.ppc, g.HW);      // multiples of 16
  const bf16* xn = x + ((size_t)n * g.HW + hp) * CB + 8 * lc;
  bf16* an = a + ((size_t)n * g.HW + hp) * CB + 8 * lc;
  bf16* sn = xs + ((size_t)n * g.HW + kg) * CS + 8 * p;               // store layout of the 16 x CS tile: pixel 4 i + kg, channels 8 p .. + 7
  int q = p0 + 16 * wv;
  if (q >= p1) return;
  bf16x8 xr[8];
#pragma unroll
  for (int r = 0; r < 8; r++) xr[r] = ldnt(xn + (size_t)(q + 2 * r) * CB);
  for (; q < p1; q += 16 * NW) {
    // the raw tile (the conv's operand) and the normalised activation, both from the same registers
#pragma unroll
    for (int r = 0; r < 8; r++) {
      *reinterpret_cast<bf16x8*>(tile + (2 * r + hp) * BROW + 16 * lc) = xr[r];
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const float t = (float)xr[r][e] * sc[e] + sh[e];
        o[e] = (bf16)(ACT == 1 ? t * sigmoidf_(t) : t);
      }
      *reinterpret_cast<bf16x8*>(an + (size_t)(q + 2 * r) * CB) = o;
    }
    {  // the next run's loads (unconditional: past the end the last run is read again -- a branch here would make the loop top wait for every store)
      const int qn = q + 16 * NW < p1 ? q + 16 * NW : q;
#pragma unroll
      for (int r = 0; r < 8; r++) xr[r] = ldnt(xn + (size_t)(qn + 2 * r) * CB);
    }
    wave_sync();
    bf16x8 bf[CB / 32];
#pragma unroll
    for (int s = 0; s < CB / 32; s++) bf[s] = *reinterpret_cast<const bf16x8*>(tile + p * BROW + 64 * s + 16 * kg);
#pragma unroll
    for (int j0 = 0; j0 < CS / 16; j0 += 4) {
      f32x4 acc[4];
#pragma unroll
      for (int jj = 0; jj < 4; jj++) acc[jj] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < CB / 32; s++)
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
          const bf16x8 af = *reinterpret_cast<const bf16x8*>(smem + (((j0 + jj) * (CB / 32) + s) * 64 + l) * 16);
          acc[jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf[s], acc[jj], 0, 0, 0);
        }
#pragma unroll
      for (int jj = 0; jj < 4; jj++) {
        const f32x4 v = acc[jj] + bs[j0 + jj];
        const bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
        *reinterpret_cast<bf16x4*>(tile + p * SROW + (16 * (j0 + jj) + 4 * kg) * 2) = o;      // over the x tile: every fragment of it is in registers by now
      }
    }
    wave_sync();
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(tile + (4 * i + kg) * SROW + 16 * p);
      *reinterpret_cast<bf16x8*>(sn + (size_t)(q + 4 * i) * CS) = v;
    }
    wave_sync();      // the tile is rewritten by the next run
  }
}

// ---- backward: dx = rstd (dy' gamma - (s1 + x_hat s2) / m) + bf16(dy_s Ws),  dy' = da act'(.) -------------------------------------------------------------
// The elementwise arithmetic is groupnorm.hip::bwd_apply_kernel's, expression for expression; COLS and the parameter gradients ride along as they do there.
template <int ACT>
__global__ __launch_bounds__(NW * 64) void short_bwd_apply_kernel(const bf16* __restrict__ da, const bf16* __restrict__ x, const bf16* __restrict__ dys,
                                                                 const bf16* __restrict__ wt, const float* __restrict__ stats, const float* __restrict__ S,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta, bf16* __restrict__ dx, Geom g,
                                                                 float* __restrict__ colpart, const float* __restrict__ AB, float* __restrict__ dgamma,
                                                                 float* __restrict__ dbeta, int N, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  fill_weights<CB, CS>(smem, wt);
  if (AB && blockIdx.x == 0 && blockIdx.y == 0) {
    for (int c = threadIdx.x; c < CB; c += NW * 64) {
      double sa = 0.0, sb = 0.0;
      int n = 0;
      for (; n + 8 <= N; n += 8) {
        f32x2 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = *reinterpret_cast<const f32x2*>(AB + ((size_t)(n + u) * CB + c) * 2);
#pragma unroll
        for (int u = 0; u < 8; u++) { sa += v[u][0]; sb += v[u][1]; }
      }
      for (; n < N; n++) { sa += AB[((size_t)n * CB + c) * 2]; sb += AB[((size_t)n * CB + c) * 2 + 1]; }
      dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)sa;
      dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)sb;
    }
  }
  const int wv = threadIdx.x >> 6, l = threadIdx.x & 63, p = l & 15, kg = l >> 4;
  const int lc = l & 31, hp = l >> 5;
  const int n = blockIdx.y;
  char* tile = smem + W_BYTES + wv * TILE;
  const float inv_m = 1.0f / ((float)g.cpg * (float)g.HW);
  float mu[8], rs[8], ga[8], be[8], s1[8], s2[8], cs[8];
#pragma unroll
  for (int e = 0; e < 8; e++) {
    const int c = 8 * lc + e;
    const size_t gi = ((size_t)n * g.G + c / g.cpg) * 2;
    mu[e] = stats[gi]; rs[e] = stats[gi + 1]; ga[e] = gamma[c]; be[e] = beta[c];
    s1[e] = S[gi] * inv_m; s2[e] = S[gi + 1] * inv_m;
    cs[e] = 0.f;
  }
  __syncthreads();
  const int p0 = blockIdx.x * g.ppc, p1 = min(p0 + g.ppc, g.HW);
  const size_t eb = ((size_t)n * g.HW + hp) * CB + 8 * lc;
  const bf16* yn = dys + ((size_t)n * g.HW + kg) * CS + 8 * p;        // load layout of the 16 x CS tile: pixel 4 i + kg, channels 8 p .. + 7
  int q = p0 + 16 * wv;
  if (q < p1) {
    bf16x8 yr[4];
#pragma unroll
    for (int i = 0; i < 4; i++) yr[i] = ldnt(yn + (size_t)(q + 4 * i) * CS);
    for (; q < p1; q += 16 * NW) {
#pragma unroll
      for (int i = 0; i < 4; i++) *reinterpret_cast<bf16x8*>(tile + (4 * i + kg) * SROW + 16 * p) = yr[i];
      bf16x8 xr[8], dr[8];
#pragma unroll
      for (int r = 0; r < 8; r++) {
        xr[r] = ldnt(x + eb + (size_t)(q + 2 * r) * CB);
        dr[r] = ldnt(da + eb + (size_t)(q + 2 * r) * CB);
      }
      {
        const int qn = q + 16 * NW < p1 ? q + 16 * NW : q;
#pragma unroll
        for (int i = 0; i < 4; i++) yr[i] = ldnt(yn + (size_t)(qn + 4 * i) * CS);
      }
      wave_sync();
      bf16x8 bf[CS / 32];
#pragma unroll
      for (int s = 0; s < CS / 32; s++) bf[s] = *reinterpret_cast<const bf16x8*>(tile + p * SROW + 64 * s + 16 * kg);
#pragma unroll
      for (int j0 = 0; j0 < CB / 16; j0 += 4) {
        f32x4 acc[4];
#pragma unroll
        for (int jj = 0; jj < 4; jj++) acc[jj] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < CS / 32; s++)
#pragma unroll
          for (int jj = 0; jj < 4; jj++) {
            const bf16x8 af = *reinterpret_cast<const bf16x8*>(smem + (((j0 + jj) * (CS / 32) + s) * 64 + l) * 16);
            acc[jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf[s], acc[jj], 0, 0, 0);
          }
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
          const bf16x4 o = {(bf16)acc[jj][0], (bf16)acc[jj][1], (bf16)acc[jj][2], (bf16)acc[jj][3]};      // the stored gradient's rounding site
          *reinterpret_cast<bf16x4*>(tile + p * BROW + (16 * (j0 + jj) + 4 * kg) * 2) = o;              // over the dy tile: its fragments are in registers
        }
      }
      wave_sync();
#pragma unroll
      for (int r = 0; r < 8; r++) {
        const bf16x8 rr = *reinterpret_cast<const bf16x8*>(tile + (2 * r + hp) * BROW + 16 * lc);
        bf16x8 ob;
#pragma unroll
        for (int e = 0; e < 8; e++) {
          const float xh = ((float)xr[r][e] - mu[e]) * rs[e];
          float dy = (float)dr[r][e];
          if (ACT == 1) {
            const float t = xh * ga[e] + be[e];
            const float sg = sigmoidf_(t);
            dy *= sg * (1.f + t * (1.f - sg));
          }
          const float rv = rs[e] * (dy * ga[e] - s1[e] - xh * s2[e]);
          ob[e] = (bf16)((float)rr[e] + rv);
          cs[e] += (float)ob[e];
        }
        *reinterpret_cast<bf16x8*>(dx + eb + (size_t)(q + 2 * r) * CB) = ob;
      }
      wave_sync();
    }
  }
  if (colpart) {      // column sums of the stored dx over this block's pixels: 16 partials per channel (wave, pixel parity), added in that order
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem + W_BYTES);
#pragma unroll
    for (int e = 0; e < 8; e++) red[(2 * wv + hp) * CB + 8 * lc + e] = cs[e];
    __syncthreads();
    if (threadIdx.x < CB) {
      float s = 0.f;
      for (int r = 0; r < 2 * NW; r++) s += red[r * CB + threadIdx.x];
      colpart[((size_t)n * g.nchunk + blockIdx.x) * CB + threadIdx.x] = s;
    }
  }
}

// ---- 512 <-> 256 channels: the weight in the waves' registers, one run per block at a time -------------------------------------------------------------------
namespace coop {
constexpr int CB = 512, CS = 256;
constexpr int BROW = CB * 2 + 16, SROW = CS * 2 + 16;     // bytes per pixel row of the 16 x CB / 16 x CS tiles
typedef int i32x4 __attribute__((ext_vector_type(4)));

// acc (+)= A (an accumulation-file register quad: the wave's weight fragment) x B
__device__ __forceinline__ void mfma_a(f32x4& acc, const i32x4& a, const bf16x8& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "v"(b));
}

template <int ACT>
__global__ __launch_bounds__(512) void coop_apply_kernel(const bf16* __restrict__ x, const float* __restrict__ stats, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const bf16* __restrict__ w, const float* __restrict__ bias,
                                                         bf16* __restrict__ a, bf16* __restrict__ xs, Geom g) {
  __shared__ __attribute__((aligned(16))) char xT[16 * BROW];
  __shared__ __attribute__((aligned(16))) char oT[16 * SROW];
  const int wv = threadIdx.x >> 6, l = threadIdx.x & 63, p = l & 15, kg = l >> 4;
  const int n = blockIdx.y;
  // this wave's output channels 32 wv .. + 31: fragments (j, s) = rows 32 wv + 16 j + (l & 15), k = 32 s + 8 kg .. + 7 of W [CS][CB]
  i32x4 wf[2 * (CB / 32)];
#pragma unroll
  for (int j = 0; j < 2; j++)
#pragma unroll
    for (int s = 0; s < CB / 32; s++)
      wf[j * (CB / 32) + s] = *reinterpret_cast<const i32x4*>(w + (size_t)(32 * wv + 16 * j + p) * CB + 32 * s + 8 * kg);
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; e++) {
    const int c = 8 * l + e;
    const float* st = stats + ((size_t)n * g.G + c / g.cpg) * 2;
    sc[e] = st[1] * gamma[c];
    sh[e] = beta[c] - st[0] * sc[e];
  }
  f32x4 bs[2];
#pragma unroll
  for (int j = 0; j < 2; j++)
#pragma unroll
    for (int i = 0; i < 4; i++) bs[j][i] = bias ? bias[32 * wv + 16 * j + 4 * kg + i] : 0.f;
  const int p0 = blockIdx.x * g.ppc, p1 = min(p0 + g.ppc, g.HW);      // multiples of 16
  const size_t eb = ((size_t)n * g.HW + wv) * CB + 8 * l;             // elementwise layout: pixels wv and wv + 8 of a run, channels 8 l .. + 7
  bf16* sn = xs + ((size_t)n * g.HW + (threadIdx.x >> 5)) * CS + 8 * (threadIdx.x & 31);      // store layout of the 16 x CS tile
  bf16x8 xn[2];
#pragma unroll
  for (int r = 0; r < 2; r++) xn[r] = ldnt(x + eb + (size_t)(p0 + 8 * r) * CB);
  for (int q = p0; q < p1; q += 16) {
    bf16x8 xr[2];
#pragma unroll
    for (int r = 0; r < 2; r++) xr[r] = xn[r];
    {  // the next run's loads, a whole run ahead (unconditional: past the end this run is read again)
      const int qn = q + 16 < p1 ? q + 16 : q;
#pragma unroll
      for (int r = 0; r < 2; r++) xn[r] = ldnt(x + eb + (size_t)(qn + 8 * r) * CB);
    }
#pragma unroll
    for (int r = 0; r < 2; r++) {
      *reinterpret_cast<bf16x8*>(xT + (wv + 8 * r) * BROW + 16 * l) = xr[r];
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const float t = (float)xr[r][e] * sc[e] + sh[e];
        o[e] = (bf16)(ACT == 1 ? t * sigmoidf_(t) : t);
      }
      *reinterpret_cast<bf16x8*>(a + eb + (size_t)(q + 8 * r) * CB) = o;
    }
    __syncthreads();      // the run's pixels are in xT (and every wave is done with oT of the run before)
    f32x4 acc[2][2];      // [j][half of K]: four independent accumulate chains
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int h = 0; h < 2; h++) acc[j][h] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s0 = 0; s0 < CB / 32; s0 += 8) {
      bf16x8 bf[8];
#pragma unroll
      for (int s = 0; s < 8; s++) bf[s] = *reinterpret_cast<const bf16x8*>(xT + p * BROW + 64 * (s0 + s) + 16 * kg);
#pragma unroll
      for (int s = 0; s < 8; s++)
#pragma unroll
        for (int j = 0; j < 2; j++) mfma_a(acc[j][s & 1], wf[j * (CB / 32) + s0 + s], bf[s]);
    }
    asm volatile("s_nop 15" ::: "memory");     // MFMA (inline asm, invisible to the hazard recognizer) -> VALU read of its result
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const f32x4 v = acc[j][0] + acc[j][1] + bs[j];
      const bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
      *reinterpret_cast<bf16x4*>(oT + p * SROW + (32 * wv + 16 * j + 4 * kg) * 2) = o;
    }
    __syncthreads();      // every wave's slice of the result is in oT (and every fragment read of xT is done)
    *reinterpret_cast<bf16x8*>(sn + (size_t)q * CS) = *reinterpret_cast<const bf16x8*>(oT + (threadIdx.x >> 5) * SROW + 16 * (threadIdx.x & 31));
  }
}

template <int ACT>
__global__ __launch_bounds__(512) void coop_bwd_apply_kernel(const bf16* __restrict__ da, const bf16* __restrict__ x, const bf16* __restrict__ dys,
                                                             const bf16* __restrict__ wt, const float* __restrict__ stats, const float* __restrict__ S,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta, bf16* __restrict__ dx, Geom g,
                                                             float* __restrict__ colpart, const float* __restrict__ AB, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, int N, int accumulate) {
  __shared__ __attribute__((aligned(16))) char yT[16 * SROW];
  __shared__ __attribute__((aligned(16))) char oT[16 * BROW];      // 16640 B: also the column sums' fold (8 x 512 floats)
  if (AB && blockIdx.x == 0 && blockIdx.y == 0) {
    for (int c = threadIdx.x; c < CB; c += 512) {
      double sa = 0.0, sb = 0.0;
      int n = 0;
      for (; n + 8 <= N; n += 8) {
        f32x2 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = *reinterpret_cast<const f32x2*>(AB + ((size_t)(n + u) * CB + c) * 2);
#pragma unroll
        for (int u = 0; u < 8; u++) { sa += v[u][0]; sb += v[u][1]; }
      }
      for (; n < N; n++) { sa += AB[((size_t)n * CB + c) * 2]; sb += AB[((size_t)n * CB + c) * 2 + 1]; }
      dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)sa;
      dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)sb;
    }
  }
  const int wv = threadIdx.x >> 6, l = threadIdx.x & 63, p = l & 15, kg = l >> 4;
  const int n = blockIdx.y;
  // this wave's output channels 64 wv .. + 63: fragments (j, s) = rows 64 wv + 16 j + (l & 15), k = 32 s + 8 kg .. + 7 of Wt [CB][CS]
  i32x4 wf[4 * (CS / 32)];
#pragma unroll
  for (int j = 0; j < 4; j++)
#pragma unroll
    for (int s = 0; s < CS / 32; s++)
      wf[j * (CS / 32) + s] = *reinterpret_cast<const i32x4*>(wt + (size_t)(64 * wv + 16 * j + p) * CS + 32 * s + 8 * kg);
  const float inv_m = 1.0f / ((float)g.cpg * (float)g.HW);
  // a lane's eight channels lie in ONE group (cpg % 8 == 0, make_geom): the per-group values are scalars here (same values, 28 registers fewer)
  const size_t gi = ((size_t)n * g.G + (8 * l) / g.cpg) * 2;
  const float mu = stats[gi], rs = stats[gi + 1], s1 = S[gi] * inv_m, s2 = S[gi + 1] * inv_m;
  float ga[8], be[8], cs[8];
#pragma unroll
  for (int e = 0; e < 8; e++) { ga[e] = gamma[8 * l + e]; be[e] = beta[8 * l + e]; cs[e] = 0.f; }
  const int p0 = blockIdx.x * g.ppc, p1 = min(p0 + g.ppc, g.HW);
  const size_t eb = ((size_t)n * g.HW + wv) * CB + 8 * l;
  const bf16* yn = dys + ((size_t)n * g.HW + (threadIdx.x >> 5)) * CS + 8 * (threadIdx.x & 31);      // load layout of the 16 x CS tile
  bf16x8 yr = ldnt(yn + (size_t)p0 * CS), xn[2], dn[2];
#pragma unroll
  for (int r = 0; r < 2; r++) { xn[r] = ldnt(x + eb + (size_t)(p0 + 8 * r) * CB); dn[r] = ldnt(da + eb + (size_t)(p0 + 8 * r) * CB); }
  for (int q = p0; q < p1; q += 16) {
    *reinterpret_cast<bf16x8*>(yT + (threadIdx.x >> 5) * SROW + 16 * (threadIdx.x & 31)) = yr;
    bf16x8 xr[2], dr[2];
#pragma unroll
    for (int r = 0; r < 2; r++) { xr[r] = xn[r]; dr[r] = dn[r]; }
    {
      const int qn = q + 16 < p1 ? q + 16 : q;
      yr = ldnt(yn + (size_t)qn * CS);
#pragma unroll
      for (int r = 0; r < 2; r++) { xn[r] = ldnt(x + eb + (size_t)(qn + 8 * r) * CB); dn[r] = ldnt(da + eb + (size_t)(qn + 8 * r) * CB); }
    }
    __syncthreads();      // the run's dy pixels are in yT (and every wave is done with oT of the run before)
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; j++) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s0 = 0; s0 < CS / 32; s0 += 4) {      // four fragments at a time: the register file is full (128 of a lane's 256 registers hold weights)
      bf16x8 bf[4];
#pragma unroll
      for (int s = 0; s < 4; s++) bf[s] = *reinterpret_cast<const bf16x8*>(yT + p * SROW + 64 * (s0 + s) + 16 * kg);
#pragma unroll
      for (int s = 0; s < 4; s++)
#pragma unroll
        for (int j = 0; j < 4; j++) mfma_a(acc[j], wf[j * (CS / 32) + s0 + s], bf[s]);
    }
    asm volatile("s_nop 15" ::: "memory");
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const bf16x4 o = {(bf16)acc[j][0], (bf16)acc[j][1], (bf16)acc[j][2], (bf16)acc[j][3]};      // the stored gradient's rounding site
      *reinterpret_cast<bf16x4*>(oT + p * BROW + (64 * wv + 16 * j + 4 * kg) * 2) = o;
    }
    __syncthreads();      // the shortcut gradient of the run is in oT (and every fragment read of yT is done)
#pragma unroll
    for (int r = 0; r < 2; r++) {
      const bf16x8 rr = *reinterpret_cast<const bf16x8*>(oT + (wv + 8 * r) * BROW + 16 * l);
      bf16x8 ob;
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const float xh = ((float)xr[r][e] - mu) * rs;
        float dy = (float)dr[r][e];
        if (ACT == 1) {
          const float t = xh * ga[e] + be[e];
          const float sg = sigmoidf_(t);
          dy *= sg * (1.f + t * (1.f - sg));
        }
        const float rv = rs * (dy * ga[e] - s1 - xh * s2);
        ob[e] = (bf16)((float)rr[e] + rv);
        cs[e] += (float)ob[e];
      }
      *reinterpret_cast<bf16x8*>(dx + eb + (size_t)(q + 8 * r) * CB) = ob;
    }
  }
  if (colpart) {      // column sums of the stored dx over this block's pixels: one partial per wave, added in wave order
    __syncthreads();
    float* red = reinterpret_cast<float*>(oT);
#pragma unroll
    for (int e = 0; e < 8; e++) red[wv * CB + 8 * l + e] = cs[e];
    __syncthreads();
    float s = 0.f;
    for (int r = 0; r < 8; r++) s += red[r * CB + threadIdx.x];
    colpart[((size_t)n * g.nchunk + blockIdx.x) * CB + threadIdx.x] = s;
  }
}
}  // namespace coop

static int make_geom(Geom& g, int n, int hw, int c, int cs, int groups) {
  const bool wave_form = c == CB && cs == CS, coop_form = c == coop::CB && cs == coop::CS;
  if (n <= 0 || hw <= 0 || hw % 16 != 0 || !(wave_form || coop_form) || groups <= 0 || c % groups != 0 || (coop_form && (c / groups) % 8 != 0) || (long long)n * hw >= (1ll << 31)) return -1;
  g.HW = hw; g.G = groups; g.cpg = c / groups;
  // one block per CU at a time (LDS): ~3 equal blocks per CU, a block's eight waves walk 128 pixels per round
  int nchunk = (768 + n - 1) / n;
  int ppc = (hw + nchunk - 1) / nchunk;
  const int round = wave_form ? 16 * NW : 16;      // pixels a block walks per round
  ppc = (ppc + round - 1) / round * round;
  g.ppc = ppc; g.nchunk = (hw + ppc - 1) / ppc;
  return 0;
}

}  // namespace dmvae_ns
using namespace dmvae_ns;

int dmvae_gn_bwd_reduce_parts(const void* da, const void* x, const void* stats, const void* gamma, const void* beta, void* workspace, size_t workspace_bytes,
                              int n, int hw, int c, int groups, int act, float** AB, float** S, hipStream_t stream);      // groupnorm.hip
int dmvae_colsum_final(const float* part, float* out, int nparts, int C, int accumulate, hipStream_t stream);             // conv_wgrad.hip

extern "C" int dmvae_groupnorm_short_supported(int n, int hw, int c, int cs, int groups) {
  Geom g;
  return make_geom(g, n, hw, c, cs, groups) == 0 ? 1 : 0;
}

extern "C" int dmvae_groupnorm_apply_short(const void* x, const void* stats, const void* gamma, const void* beta, const void* w, const void* bias, void* a, void* xs,
                                           int n, int hw, int c, int cs, int groups, int act, hipStream_t stream) {
  Geom g;
  DMVAE_CHECK_ARG(x && stats && gamma && beta && w && a && xs, "groupnorm_apply_short: null pointer");
  DMVAE_CHECK_ARG(act == 0 || act == 1, "groupnorm_apply_short: act must be 0 (none) or 1 (swish)");
  DMVAE_CHECK_ARG(make_geom(g, n, hw, c, cs, groups) == 0, "groupnorm_apply_short: unsupported shape n=%d hw=%d c=%d cs=%d groups=%d (c / cs = 256 / 128 or 512 / 256, hw %% 16 == 0)",
                  n, hw, c, cs, groups);
  const dim3 grid(g.nchunk, n);
  if (c == coop::CB) {
    if (act == 1) hipLaunchKernelGGL(coop::coop_apply_kernel<1>, grid, dim3(512), 0, stream, (const bf16*)x, (const float*)stats, (const float*)gamma, (const float*)beta,
                                     (const bf16*)w, (const float*)bias, (bf16*)a, (bf16*)xs, g);
    else hipLaunchKernelGGL(coop::coop_apply_kernel<0>, grid, dim3(512), 0, stream, (const bf16*)x, (const float*)stats, (const float*)gamma, (const float*)beta,
                            (const bf16*)w, (const float*)bias, (bf16*)a, (bf16*)xs, g);
    DMVAE_CHECK_LAUNCH();
    return 0;
  }
  static bool attr[2] = {false, false};
#define DMVAE_NS_APPLY(A)                                                                                                                                   \
  do {                                                                                                                                                      \
    if (!attr[A]) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(short_apply_kernel<A>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES); attr[A] = true; } \
    hipLaunchKernelGGL(short_apply_kernel<A>, grid, dim3(NW * 64), LDS_BYTES, stream, (const bf16*)x, (const float*)stats, (const float*)gamma, (const float*)beta, \
                       (const bf16*)w, (const float*)bias, (bf16*)a, (bf16*)xs, g);                                                                         \
  } while (0)
  if (act == 1) DMVAE_NS_APPLY(1); else DMVAE_NS_APPLY(0);
#undef DMVAE_NS_APPLY
  DMVAE_CHECK_LAUNCH();
  return 0;
}

extern "C" size_t dmvae_groupnorm_bwd_short_workspace(int n, int hw, int c, int cs, int groups) {
  Geom g;
  if (make_geom(g, n, hw, c, cs, groups)) return 0;
  return dmvae_groupnorm_workspace(n, hw, c, groups) + (size_t)n * g.nchunk * c * sizeof(float);
}

extern "C" int dmvae_groupnorm_bwd_short(const void* da, const void* x, const void* dys, const void* wt, const void* stats, const void* gamma, const void* beta,
                                         void* dx, void* dgamma, void* dbeta, void* colsum, void* workspace, size_t workspace_bytes, int n, int hw, int c, int cs,
                                         int groups, int act, int accumulate, int colsum_accumulate, hipStream_t stream) {
  Geom g;
  DMVAE_CHECK_ARG(da && x && dys && wt && stats && gamma && beta && dx && workspace, "groupnorm_bwd_short: null pointer");
  DMVAE_CHECK_ARG(act == 0 || act == 1, "groupnorm_bwd_short: act must be 0 (none) or 1 (swish)");
  DMVAE_CHECK_ARG(make_geom(g, n, hw, c, cs, groups) == 0, "groupnorm_bwd_short: unsupported shape n=%d hw=%d c=%d cs=%d groups=%d (c / cs = 256 / 128 or 512 / 256, hw %% 16 == 0)",
                  n, hw, c, cs, groups);
  DMVAE_CHECK_ARG((dgamma == nullptr) == (dbeta == nullptr), "groupnorm_bwd_short: dgamma and dbeta go together");
  DMVAE_CHECK_ARG(workspace_bytes >= dmvae_groupnorm_bwd_short_workspace(n, hw, c, cs, groups), "groupnorm_bwd_short: workspace too small");
  const size_t base = dmvae_groupnorm_workspace(n, hw, c, groups);
  float *AB = nullptr, *S = nullptr;
  int rc = dmvae_gn_bwd_reduce_parts(da, x, stats, gamma, beta, workspace, base, n, hw, c, groups, act, &AB, &S, stream);
  if (rc) return rc;
  float* colpart = colsum ? reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + base) : nullptr;
  const dim3 grid(g.nchunk, n);
  if (c == coop::CB) {
#define DMVAE_NS_CBWD(A)                                                                                                                                        \
    hipLaunchKernelGGL(coop::coop_bwd_apply_kernel<A>, grid, dim3(512), 0, stream, (const bf16*)da, (const bf16*)x, (const bf16*)dys, (const bf16*)wt,            \
                       (const float*)stats, (const float*)S, (const float*)gamma, (const float*)beta, (bf16*)dx, g, colpart,                                    \
                       dgamma ? (const float*)AB : (const float*)nullptr, (float*)dgamma, (float*)dbeta, n, accumulate)
    if (act == 1) DMVAE_NS_CBWD(1); else DMVAE_NS_CBWD(0);
#undef DMVAE_NS_CBWD
    DMVAE_CHECK_LAUNCH();
    if (colsum) return dmvae_colsum_final(colpart, (float*)colsum, n * g.nchunk, c, colsum_accumulate, stream);
    return 0;
  }
  static bool attr[2] = {false, false};
#define DMVAE_NS_BWD(A)                                                                                                                                         \
  do {                                                                                                                                                          \
    if (!attr[A]) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(short_bwd_apply_kernel<A>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES); attr[A] = true; } \
    hipLaunchKernelGGL(short_bwd_apply_kernel<A>, grid, dim3(NW * 64), LDS_BYTES, stream, (const bf16*)da, (const bf16*)x, (const bf16*)dys, (const bf16*)wt,    \
                       (const float*)stats, (const float*)S, (const float*)gamma, (const float*)beta, (bf16*)dx, g, colpart,                                    \
                       dgamma ? (const float*)AB : (const float*)nullptr, (float*)dgamma, (float*)dbeta, n, accumulate);                                        \
  } while (0)
  if (act == 1) DMVAE_NS_BWD(1); else DMVAE_NS_BWD(0);
#undef DMVAE_NS_BWD
  DMVAE_CHECK_LAUNCH();
  if (colsum) return dmvae_colsum_final(colpart, (float*)colsum, n * g.nchunk, c, colsum_accumulate, stream);
  return 0;
}
