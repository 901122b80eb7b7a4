// Elementwise kernels of the frozen ViT encoder forward (models/vae.py:52-53 -> timm VisionTransformer blocks; block algebra as
// in the reference's vendored models/dino_layers/block.py:89-115): the f32 residual stream stays in HBM, everything that
// feeds a GEMM is bf16.  Both are HBM-bound row kernels (16-B accesses, one wave per row, f32 statistics).
//   layernorm_f32_bf16 : y = bf16( (x - mean) * rstd * gamma + beta )      replaces nn.LayerNorm (f32 under autocast) + the bf16 cast
//   scale_residual_f32 : x += gamma * float(y)                              replaces LayerScale (dino_layers/layer_scale.py:15-26) + the
//                                                                          residual add (f32 under autocast type promotion)
#include "common.h"
#include "dmvae_hip.h"

namespace dmvae_vit {

// one wave per row; C % 256 == 0 (4 floats per lane per sweep), C <= 4096.
// RES: the LayerScale + residual add that precedes every LayerNorm but the first (x += ls * r, the previous branch's output r in bf16) in the same pass --
// the f32 residual stream is read and written once instead of read-written by one kernel and read again by the next (bit-identical to the two kernels).
template <int SWEEPS, bool RES = false>
__global__ __launch_bounds__(256) void layernorm_kernel(float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, bf16* __restrict__ y, int rows, float eps,
                                                        const bf16* __restrict__ r = nullptr, const float* __restrict__ ls = nullptr) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  constexpr int C = SWEEPS * 256;
  float* xr = x + (size_t)row * C;
  f32x4 v[SWEEPS];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < SWEEPS; k++) {
    v[k] = *reinterpret_cast<const f32x4*>(xr + k * 256 + lane * 4);
    if constexpr (RES) {
      const bf16x4 rv = *reinterpret_cast<const bf16x4*>(r + (size_t)row * C + k * 256 + lane * 4);
      const f32x4 g = *reinterpret_cast<const f32x4*>(ls + k * 256 + lane * 4);
#pragma unroll
      for (int e = 0; e < 4; e++) v[k][e] = fmaf(g[e], (float)rv[e], v[k][e]);
      *reinterpret_cast<f32x4*>(xr + k * 256 + lane * 4) = v[k];
    }
    s += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
  }
  const float mean = wave_sum(s) * (1.f / C);
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < SWEEPS; k++)
#pragma unroll
    for (int e = 0; e < 4; e++) { const float d = v[k][e] - mean; ss += d * d; }
  const float rstd = rsqrtf(wave_sum(ss) * (1.f / C) + eps);
  bf16* yr = y + (size_t)row * C;
#pragma unroll
  for (int k = 0; k < SWEEPS; k++) {
    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + k * 256 + lane * 4);
    const f32x4 b = *reinterpret_cast<const f32x4*>(beta + k * 256 + lane * 4);
    bf16x4 o;
#pragma unroll
    for (int e = 0; e < 4; e++) o[e] = (bf16)((v[k][e] - mean) * rstd * g[e] + b[e]);
    *reinterpret_cast<bf16x4*>(yr + k * 256 + lane * 4) = o;
  }
}

__global__ __launch_bounds__(256) void scale_residual_kernel(float* __restrict__ x, const bf16* __restrict__ y, const float* __restrict__ gamma,
                                                             size_t n8, int c8) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % c8) * 8;
    const bf16x8 v = reinterpret_cast<const bf16x8*>(y)[i];
    f32x4 a = reinterpret_cast<const f32x4*>(x)[2 * i], b = reinterpret_cast<const f32x4*>(x)[2 * i + 1];
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + c), g1 = *reinterpret_cast<const f32x4*>(gamma + c + 4);
#pragma unroll
    for (int e = 0; e < 4; e++) { a[e] = fmaf(g0[e], (float)v[e], a[e]); b[e] = fmaf(g1[e], (float)v[4 + e], b[e]); }
    reinterpret_cast<f32x4*>(x)[2 * i] = a;
    reinterpret_cast<f32x4*>(x)[2 * i + 1] = b;
  }
}

// P[r][:] = softmax(scale * S[r][:]) for bf16 scores (the rounded QK^T of the encoder's attention), f32 inside, bf16 out; one wave
// per row, cols <= 512.  Replaces the scale multiply, the f32 up-cast, softmax and the bf16 down-cast (four ATen kernels).
__global__ __launch_bounds__(256) void softmax_bf16_kernel(const bf16* __restrict__ s, bf16* __restrict__ p, size_t rows, int cols, float scale) {
  const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const bf16* sr = s + row * cols;
  float v[8];
  float m = -INFINITY;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int c = lane + k * 64;
    v[k] = c < cols ? (float)sr[c] * scale : -INFINITY;
    m = fmaxf(m, v[k]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < 8; k++) { v[k] = __expf(v[k] - m); sum += v[k]; }
  const float inv = 1.f / wave_sum(sum);
  bf16* pr = p + row * cols;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int c = lane + k * 64;
    if (c < cols) pr[c] = (bf16)(v[k] * inv);
  }
}

}  // namespace dmvae_vit
using namespace dmvae_vit;

extern "C" int dmvae_layernorm_f32_bf16(const void* x, const void* gamma, const void* beta, void* y, int rows, int c, float eps,
                                        hipStream_t stream) {
  DMVAE_CHECK_ARG(x && gamma && beta && y && rows > 0, "layernorm_f32_bf16: bad argument");
  DMVAE_CHECK_ARG(c == 256 || c == 512 || c == 768 || c == 1024 || c == 1280 || c == 1536,
                  "layernorm_f32_bf16: width must be a multiple of 256 up to 1536 (got %d)", c);
  const dim3 grid((rows + 3) / 4), block(256);
  switch (c / 256) {
    case 1: hipLaunchKernelGGL(layernorm_kernel<1>, grid, block, 0, stream, (float*)x, (const float*)gamma, (const float*)beta, (bf16*)y, rows, eps, nullptr, nullptr); break;
    case 2: hipLaunchKernelGGL(layernorm_kernel<2>, grid, block, 0, stream, (float*)x, (const float*)gamma, (const float*)beta, (bf16*)y, rows, eps, nullptr, nullptr); break;
    case 3: hipLaunchKernelGGL(layernorm_kernel<3>, grid, block, 0, stream, (float*)x, (const float*)gamma, (const float*)beta, (bf16*)y, rows, eps, nullptr, nullptr); break;
    case 4: hipLaunchKernelGGL(layernorm_kernel<4>, grid, block, 0, stream, (float*)x, (const float*)gamma, (const float*)beta, (bf16*)y, rows, eps, nullptr, nullptr); break;
    case 5: hipLaunchKernelGGL(layernorm_kernel<5>, grid, block, 0, stream, (float*)x, (const float*)gamma, (const float*)beta, (bf16*)y, rows, eps, nullptr, nullptr); break;
    default: hipLaunchKernelGGL(layernorm_kernel<6>, grid, block, 0, stream, (float*)x, (const float*)gamma, (const float*)beta, (bf16*)y, rows, eps, nullptr, nullptr); break;
  }
  DMVAE_CHECK_LAUNCH();
  return 0;
}

extern "C" int dmvae_scale_residual_layernorm(void* x, const void* r, const void* ls_gamma, const void* gamma, const void* beta, void* y, int rows, int c,
                                              float eps, hipStream_t stream) {
  DMVAE_CHECK_ARG(x && r && ls_gamma && gamma && beta && y && rows > 0, "scale_residual_layernorm: bad argument");
  DMVAE_CHECK_ARG(c == 256 || c == 512 || c == 768 || c == 1024 || c == 1280 || c == 1536,
                  "scale_residual_layernorm: width must be a multiple of 256 up to 1536 (got %d)", c);
  const dim3 grid((rows + 3) / 4), block(256);
#define DMVAE_SRLN(S) hipLaunchKernelGGL((layernorm_kernel<S, true>), grid, block, 0, stream, (float*)x, (const float*)gamma, (const float*)beta, (bf16*)y, rows, eps, \
                                         (const bf16*)r, (const float*)ls_gamma)
  switch (c / 256) {
    case 1: DMVAE_SRLN(1); break;
    case 2: DMVAE_SRLN(2); break;
    case 3: DMVAE_SRLN(3); break;
    case 4: DMVAE_SRLN(4); break;
    case 5: DMVAE_SRLN(5); break;
    default: DMVAE_SRLN(6); break;
  }
#undef DMVAE_SRLN
  DMVAE_CHECK_LAUNCH();
  return 0;
}

extern "C" int dmvae_scale_residual_f32(void* x, const void* y, const void* gamma, size_t rows, int c, hipStream_t stream) {
  DMVAE_CHECK_ARG(x && y && gamma && c > 0 && c % 8 == 0, "scale_residual_f32: width must be a multiple of 8");
  if (rows == 0) return 0;
  const size_t n8 = rows * (size_t)(c / 8);
  size_t nb = (n8 + 255) / 256; if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(scale_residual_kernel, dim3((unsigned)nb), dim3(256), 0, stream, (float*)x, (const bf16*)y, (const float*)gamma, n8, c / 8);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

extern "C" int dmvae_softmax_rows_bf16(const void* s, void* p, size_t rows, int cols, float scale, hipStream_t stream) {
  DMVAE_CHECK_ARG(s && p && cols > 0 && cols <= 512, "softmax_rows_bf16: cols must be in 1..512 (got %d)", cols);
  if (rows == 0) return 0;
  hipLaunchKernelGGL(softmax_bf16_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, (const bf16*)s, (bf16*)p, rows, cols, scale);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

// ---- fused multi-head self-attention of the encoder (timm Attention / dino_layers/attention.py:56-69; head dim 64, S <= 288) ------
// qkv: [B][S][3][H][64] bf16 (the qkv Linear's output as it lies in memory), out: [B][S][H*64] bf16.
// One workgroup per (batch, head): K ([key][d], 128- / 256-B rows, 16-B chunks XOR-swizzled per key: kslot) and V ([key][128-slot rows] with
// the 64-B segment swizzle of the wgrad kernels, so the same ds_read_b64_tr_b16 addressing applies) are staged in LDS once; each
// wave walks 32-query blocks:  S^T = K Q^T on the matrix cores (so a lane owns one query column and softmax needs no
// cross-lane traffic beyond one swap with lane^32), f32 softmax with the 1/sum folded into P, P -> bf16 A-fragments by
// v_permlane32_swap, O = P V with V fragments from the LDS transpose read.  Nothing of size S x S ever reaches HBM.
namespace dmvae_vit {

constexpr int ATT_D = 64, ATT_KB = 9, ATT_KEYS = ATT_KB * 32;  // keys padded to 288

template <int N>
__device__ __forceinline__ void attn_wait_vmcnt() {   // through the builtin: the compiler's wait-count pass has to see the wait (conv_pp.hip)
  __builtin_amdgcn_s_waitcnt((N & 15) | 0x0F70 | ((N >> 4) << 14));
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ s16x4 tr_read_v(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  bf16x2 t = {(bf16)a, (bf16)b};
  return *reinterpret_cast<unsigned*>(&t);
}

// DP: head dim as staged (64, or 96 = 72 zero-padded by the producer for LightningDiT-XL); D: real head dim (V / output width).
// q / k / v: per-(batch, head) base = ptr + b * bs + h * hs (elements), token rows `rs` elements apart.
struct AttnArgs {
  const bf16 *q, *k, *v;
  bf16* out;
  long long q_bs, q_hs, k_bs, k_hs, v_bs, v_hs;
  int q_rs, k_rs, v_rs;
  int S, H, D;
  int QD;          // channels a q / k row HOLDS (non-NR forms): the staged width DP (rows zero-padded by the producer), or D itself -- rows of 72 channels, 144 B apart, whose
                   // channels D .. DP - 1 are zeros by construction (a chunk past QD is not loaded); NR forms read rows of D channels out of the packed qkv
  float scale;
  int BH;          // batch * heads: blocks past it are the single-query blocks (eight (batch, head) pairs each)
  int xcd;         // 1: block -> (batch, head) through xcd_remap, so that the blocks resident on one XCD are CONSECUTIVE heads of the same samples and the 128-B lines
                   // their head slices share (a 72-channel head is 144 B of a packed qkv row) are fetched into that XCD's L2 once (DMVAE_ATTN_XCD=0: plain order)
  // NR variant: per-head RMSNorm (bf16 result) * weight and the 2-D rotary embedding are applied to q and k on their way in
  const float *qw, *kw, *cosb, *sinb;   // [D], [D], [S][D], [S][D]
  float eps;
  float* lse;      // optional [B * H][S] f32: scale * max + log(sum) of every query's scaled scores -- what the backward kernel needs to rebuild P without a pass of its own
};

// NT threads: 512 = two waves per SIMD.  One wave's softmax (the kernel's VALU-bound part: 144 exponentials and their bookkeeping per lane and query block)
// then runs under the other's matrix work and memory latency; with 256 threads the query blocks of a head were three serial rounds of load -> MFMA ->
// softmax -> MFMA -> store per wave.  
constexpr int ATTN_THREADS = 512;
// (Serving the class token's query -- S = 32 k + 1 -- by extra single-query workgroups instead of a ninth 32-query block was built and measured slower: DESIGN_HISTORY.md 9.8-6.)
// PIPE (DP = 96 route, more than one (batch, head) per workgroup): the grid is one workgroup per CU and a workgroup walks (batch, head) items blockIdx.x, + gridDim.x, ...;
// the NEXT item's K / V global loads are issued into registers right behind the current item's Q loads and land under its two sweeps -- the staging phase (a third of an
// item's time with one 147-KB workgroup per CU and nothing else resident to hide it) then only pays its LDS stores.  Same operations on the same values: same bits.
template <int DP, bool NR, bool PIPE = false>
__global__ __launch_bounds__(ATTN_THREADS) void attention_kernel(AttnArgs a) {
  constexpr int NT = ATTN_THREADS;
#if __HIP_DEVICE_COMPILE__
  constexpr int KROW = DP == 64 ? 128 : 256;   // bytes per K row in LDS (8 or 16 chunks of 16 B, XOR-swizzled per key: kslot)
  // A ds_read_b128 is served in four groups of sixteen lanes that are NOT consecutive ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... -- MI355X_MICROARCH.md, LDS
  // table), each lane reading key (lane & 31)'s chunk: the sixteen slots of a group must be distinct mod 256 B.  256-B rows: XOR with key & 15 (sixteen distinct
  // values in every group).  128-B rows (two keys per 256 B): XOR with (key >> 1) & 7 -- eight values, each met by one even and one odd key of the group.  The
  // first form of this kernel XOR-ed key & 7, which every group holds twice: a 2-way conflict on every K read (SQ_LDS_BANK_CONFLICT 37 % of the LDS cycles).
  auto kslot = [](int key, int c) { return key * KROW + ((c ^ (KROW == 128 ? (key >> 1) & 7 : key & 15)) << 4); };
  constexpr int KSTEPS = DP / 16, DB = DP / 32;
  // V rows: DP = 64 -> 128 B (two 64-B segments, swizzled by (key >> 1) & 1: the four key rows a transpose-read pass touches then sit in four distinct 64-B bank
  // slots of the 256-B LDS row); wider heads -> 256 B (four segments, swizzled by key & 3).  72 KiB per workgroup at DP = 64: TWO workgroups per CU -- with
  // S = 257 a head has nine 32-query blocks for eight waves, so one wave works a second round while seven idle, and a second resident head fills those slots.
  constexpr int VROW = DP == 64 ? 128 : 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ks = smem;                       // [288][KROW]
  char* vs = smem + ATT_KEYS * KROW;     // [288][VROW]
  auto vslot = [](int key, int c) { return VROW == 128 ? key * 128 + ((((c >> 2) ^ ((key >> 1) & 1))) << 6) + ((c & 3) << 4)
                                                        : key * 256 + ((((c >> 2) ^ (key & 3))) << 6) + ((c & 3) << 4); };
  const int S = a.S, H = a.H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Sq = S;      // queries the 32-query blocks cover
  static_assert(!(PIPE && NR), "PIPE: the plain staging only");
  const int vchunks = a.D / 8;           // V rows hold the real head dim
  // item -> (batch, head): PIPE walks items blockIdx.x + i gridDim.x (gridDim.x a multiple of 8: an item stays on its workgroup's XCD), else one item per workgroup
  auto item_bh = [&](int item) { return a.xcd ? (int)xcd_remap((unsigned)item, PIPE ? (unsigned)a.BH : gridDim.x) : item; };
  constexpr int SWEEPS = (ATT_KEYS * (DP / 8) + NT - 1) / NT;
  [[maybe_unused]] uint4 kv[SWEEPS], vv[SWEEPS];
  // every global load of the staging is issued before the first LDS store (the loop form waited for each sweep's loads before issuing the next sweep's:
  // nine to fourteen serial memory round trips, a third of the kernel's time at these sizes)
  // PIPE: buffer loads -- the per-lane byte offsets are the same for every item (kept in SWEEPS + SWEEPS registers instead of a 64-bit address per load), the item's base
  // sits in the wave-uniform descriptor, and a masked element is an offset past the descriptor's range (reads zeros: no select behind the load)
  [[maybe_unused]] unsigned voK[SWEEPS], voV[SWEEPS];
  if constexpr (PIPE) {
#pragma unroll
    for (int it = 0; it < SWEEPS; it++) {
      const int i = tid + it * NT, key = i / (DP / 8), c = i - key * (DP / 8);
      const bool ok = i < ATT_KEYS * (DP / 8) && key < S;
      voK[it] = ok && c < a.QD / 8 ? (unsigned)key * (unsigned)a.k_rs * 2u + (unsigned)c * 16u : 0x80000000u;
      voV[it] = ok && c < vchunks ? (unsigned)key * (unsigned)a.v_rs * 2u + (unsigned)c * 16u : 0x80000000u;
    }
  }
  auto load_kv = [&](int bh_) {
    const int b_ = __builtin_amdgcn_readfirstlane(bh_ / H), h_ = __builtin_amdgcn_readfirstlane(bh_ % H);
    const bf16* kp = a.k + b_ * a.k_bs + h_ * a.k_hs;
    const bf16* vp = a.v + b_ * a.v_bs + h_ * a.v_hs;
    if constexpr (PIPE) {
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      const __amdgpu_buffer_rsrc_t rK = __builtin_amdgcn_make_buffer_rsrc((void*)kp, 0, 0x7fffffff, 0x00020000);
      const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc((void*)vp, 0, 0x7fffffff, 0x00020000);
#pragma unroll
      for (int it = 0; it < SWEEPS; it++) {
        const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(rK, voK[it], 0, 0), y = __builtin_amdgcn_raw_buffer_load_b128(rV, voV[it], 0, 0);
        kv[it] = uint4{x[0], x[1], x[2], x[3]}; vv[it] = uint4{y[0], y[1], y[2], y[3]};
      }
      return;
    }
#pragma unroll
    for (int it = 0; it < SWEEPS; it++) {
      const int i = tid + it * NT, key = i / (DP / 8), c = i - key * (DP / 8);
      kv[it] = uint4{0, 0, 0, 0}; vv[it] = uint4{0, 0, 0, 0};
      if (i < ATT_KEYS * (DP / 8) && key < S) {
        if (c < a.QD / 8) kv[it] = *reinterpret_cast<const uint4*>(kp + (size_t)key * a.k_rs + c * 8);
        if (c < vchunks) vv[it] = *reinterpret_cast<const uint4*>(vp + (size_t)key * a.v_rs + c * 8);
      }
    }
  };
  int item = (int)blockIdx.x;
  if constexpr (PIPE) load_kv(item_bh(item));
  for (;;) {
  const int bh = item_bh(item);
  const int b = bh / H, h = bh % H;
  const bf16* qb_ = a.q + b * a.q_bs + h * a.q_hs;
  [[maybe_unused]] const bf16* kb_ = a.k + b * a.k_bs + h * a.k_hs;
  [[maybe_unused]] const bf16* vb_ = a.v + b * a.v_bs + h * a.v_hs;
  const int next = item + (int)gridDim.x;
  [[maybe_unused]] bool pre = PIPE && next < a.BH;     // the next item's K / V loads are still to be issued
  // ---- stage K and V: DP/8 lanes x 16 B per key row ------------------------------------------------------------------------------------
  if constexpr (NR) {  // 16 lanes per key row (the first DP/8 carry data) so that the row's sum of squares is a 16-lane butterfly
    const int c = tid & 15;
    for (int key = tid >> 4; key < ATT_KEYS; key += NT / 16) {
      uint4 kv = {0, 0, 0, 0}, vv = {0, 0, 0, 0};
      const bool live = key < S && c < vchunks;
      if (live) {
        kv = *reinterpret_cast<const uint4*>(kb_ + (size_t)key * a.k_rs + c * 8);
        vv = *reinterpret_cast<const uint4*>(vb_ + (size_t)key * a.v_rs + c * 8);
      }
      float ss = dmvae_sumsq8(kv);
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) ss += __shfl_xor(ss, o, 64);
      if (live) kv = dmvae_norm_rope8(kv, rsqrtf(ss / (float)a.D + a.eps), a.kw, a.cosb, a.sinb, key, a.D, c * 8);
      if (c < DP / 8) {
        *reinterpret_cast<uint4*>(ks + kslot(key, c)) = kv;
        *reinterpret_cast<uint4*>(vs + vslot(key, c)) = vv;
      }
    }
  } else {
    if constexpr (!PIPE) load_kv(bh);
#pragma unroll
    for (int it = 0; it < SWEEPS; it++) {
      const int i = tid + it * NT, key = i / (DP / 8), c = i - key * (DP / 8);
      if (i < ATT_KEYS * (DP / 8)) {
        *reinterpret_cast<uint4*>(ks + kslot(key, c)) = kv[it];
        // V: channel chunk c (8 channels) -> 64-B segment c >> 2, swizzled per key (vslot); 16-B slot c & 3 inside it
        *reinterpret_cast<uint4*>(vs + vslot(key, c)) = vv[it];
      }
    }
  }
  __syncthreads();
  const int kg = lane >> 5, ql = lane & 31;
  // V transpose-read addressing (see conv_wgrad_pp.hip): lane supplies 4 channels of one key row
  const int g16 = (lane >> 4) & 1, rr = (lane & 15) >> 2, qq = lane & 3;
  int voff[DB];
#pragma unroll
  for (int db = 0; db < DB; db++) {
    const int ch = db * 32 + 16 * g16 + 4 * qq;
    voff[db] = (kg * 8 + rr) * VROW + ((((ch >> 5) ^ (VROW == 128 ? (rr >> 1) & 1 : rr))) << 6) + (ch & 31) * 2;
  }
  for (int qb = wave; qb * 32 < Sq; qb += NT / 64) {
    const int q = qb * 32 + ql;
    // Q fragments (B operand of the swapped product): 8 d's per lane per 16-step
    bf16x8 qf[KSTEPS];
#pragma unroll
    for (int kk = 0; kk < KSTEPS; kk++) {
      uint4 t = {0, 0, 0, 0};
      if (q < Sq && kk * 16 + kg * 8 < (NR ? a.D : a.QD)) t = *reinterpret_cast<const uint4*>(qb_ + (size_t)q * a.q_rs + kk * 16 + kg * 8);
      qf[kk] = *reinterpret_cast<bf16x8*>(&t);
    }
    if constexpr (PIPE) {
      // behind this block's Q loads in the (in-order) load queue: waiting for Q leaves them in flight.  The waits are spelled out on both paths: left to the compiler's
      // wait-count pass, the first use of Q -- inside the sweep loop -- got s_waitcnt vmcnt(0), i.e. the prefetch was waited for before the first product
      // (and the fragments pass through an empty asm right behind the wait: the pass then knows them landed; the explicit wait alone did not change its in-loop wait)
      auto pin_q = [&]() {
#pragma unroll
        for (int kk = 0; kk < KSTEPS; kk++) asm volatile("" : "+v"(qf[kk]));
      };
      if (pre) { load_kv(item_bh(next)); pre = false; attn_wait_vmcnt<2 * SWEEPS>(); pin_q(); }
      else { attn_wait_vmcnt<0>(); pin_q(); }
    }
    if constexpr (NR) {  // this lane and lane ^ 32 hold the two halves of query q's row
      float ss = 0.f;
#pragma unroll
      for (int kk = 0; kk < KSTEPS; kk++) ss += dmvae_sumsq8(*reinterpret_cast<const uint4*>(&qf[kk]));
      ss += __shfl_xor(ss, 32, 64);
      const float rq = rsqrtf(ss / (float)a.D + a.eps);
#pragma unroll
      for (int kk = 0; kk < KSTEPS; kk++)
        if (q < S && kk * 16 + kg * 8 < a.D) {
          const uint4 o = dmvae_norm_rope8(*reinterpret_cast<const uint4*>(&qf[kk]), rq, a.qw, a.cosb, a.sinb, q, a.D, kk * 16 + kg * 8);
          qf[kk] = *reinterpret_cast<const bf16x8*>(&o);
        }
    }
    // ---- two sweeps over the key blocks, 16 score registers live instead of 144 (two waves per SIMD fit) ---------------------------------------------
    // sweep 1: S^T = K Q^T block by block for the row maximum only; sweep 2: the same product again, e = exp(s - max) straight into bf16 A fragments
    // (v_permlane32_swap), O += e V, and the row sum; O is normalised at the end.  The second QK^T costs 4-6 MFMAs per block -- the kernel is bound by the
    // exponentials, which are computed once either way.  st[r] = score(key = kb*32 + (r&3) + 8*(r>>2) + 4*kg, query q).
    auto scores = [&](int kb) {
      f32x16 st;
#pragma unroll
      for (int r = 0; r < 16; r++) st[r] = 0.f;
      const int key = kb * 32 + ql;
      bf16x8 kf[KSTEPS];      // every fragment read of the block ahead of its products (attention_bwd.hip: left to the scheduler, each product sat behind its own LDS trip)
#pragma unroll
      for (int kk = 0; kk < KSTEPS; kk++) kf[kk] = *reinterpret_cast<const bf16x8*>(ks + kslot(key, kk * 2 + kg));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kk = 0; kk < KSTEPS; kk++) st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kk], qf[kk], st, 0, 0, 0);
      // raw scores: the scale is folded into the exponential's argument below; only a block that reaches past the last key needs the mask
      if (kb * 32 + 32 > S) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int key_r = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
          st[r] = key_r < S ? st[r] : -INFINITY;
        }
      }
      return st;
    };
    const int nkb = (S + 31) >> 5;
    float m = -INFINITY;
    for (int kb = 0; kb < nkb; kb++) {
      const f32x16 st = scores(kb);
#pragma unroll
      for (int r = 0; r < 16; r++) m = fmaxf(m, st[r]);
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const float ec = a.scale * 1.4426950408889634f, emc = m * ec;      // exp(scale * (s - m)) = 2^(s * ec - m * ec): one fma + v_exp_f32 per score (scale > 0)
    float sum = 0.f;
    f32x16 o[DB];
#pragma unroll
    for (int db = 0; db < DB; db++)
#pragma unroll
      for (int r = 0; r < 16; r++) o[db][r] = 0.f;
    for (int kb = 0; kb < nkb; kb++) {
      f32x16 st = scores(kb);
      // the block's V^T fragments (both 16-key steps) are on their way while the exponentials run
      union { bf16x8 v; s16x4 hlf[2]; } vf[2][DB];
#pragma unroll
      for (int half = 0; half < 2; half++)
#pragma unroll
        for (int db = 0; db < DB; db++) {
          vf[half][db].hlf[0] = tr_read_v(vs + (kb * 2 + half) * (16 * VROW) + voff[db]);
          vf[half][db].hlf[1] = tr_read_v(vs + (kb * 2 + half) * (16 * VROW) + voff[db] + 4 * VROW);
        }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 16; r++) { st[r] = __builtin_amdgcn_exp2f(fmaf(st[r], ec, -emc)); sum += st[r]; }
#pragma unroll
      for (int half = 0; half < 2; half++) {  // 16-key step: registers r = half*8 .. half*8+7 of this block
        unsigned p0 = pack_bf16(st[half * 8 + 0], st[half * 8 + 1]);
        unsigned p1 = pack_bf16(st[half * 8 + 2], st[half * 8 + 3]);
        unsigned p2 = pack_bf16(st[half * 8 + 4], st[half * 8 + 5]);
        unsigned p3 = pack_bf16(st[half * 8 + 6], st[half * 8 + 7]);
        // lanes < 32 hold keys {0-3, 8-11} of the step, lanes >= 32 {4-7, 12-15}: the A fragment wants {0-7} / {8-15}
        auto s0 = __builtin_amdgcn_permlane32_swap(p0, p2, false, false);
        auto s1 = __builtin_amdgcn_permlane32_swap(p1, p3, false, false);
        union { unsigned u[4]; bf16x8 v; } pa;
        pa.u[0] = s0[0]; pa.u[1] = s1[0]; pa.u[2] = s0[1]; pa.u[3] = s1[1];
        // O^T = V^T P^T: the V fragment as the row operand (a lane's 8 keys of channel d are the same registers either way), so that the accumulators hold
        // channels along the registers and ONE query per lane: the row's 1 / sum is the lane's own value and a lane stores 4 consecutive channels at a time
#pragma unroll
        for (int db = 0; db < DB; db++) o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[half][db].v, pa.v, o[db], 0, 0, 0);
      }
    }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;      // both halves of the wave hold query (lane & 31)'s sum
    if (a.lse && kg == 0 && q < Sq) a.lse[(size_t)bh * S + q] = m * a.scale + __logf(sum);
    // ---- store [B][S][H*D]: lane = query q = qb*32 + (lane & 31); registers r = 4 r4 .. 4 r4 + 3 are channels db*32 + 8 r4 + 4 kg + 0..3: 8-byte stores ----
    const int C = H * a.D;
    if (q < Sq) {
      bf16* orow = a.out + ((size_t)b * S + q) * C + h * a.D;
#pragma unroll
      for (int db = 0; db < DB; db++)
#pragma unroll
        for (int r4 = 0; r4 < 4; r4++) {
          const int d0 = db * 32 + 8 * r4 + 4 * kg;
          if (d0 < a.D) {      // D % 8 == 0: the four channels are inside together
            uint2 pk;
            pk.x = pack_bf16(o[db][4 * r4 + 0] * inv, o[db][4 * r4 + 1] * inv);
            pk.y = pack_bf16(o[db][4 * r4 + 2] * inv, o[db][4 * r4 + 3] * inv);
            *reinterpret_cast<uint2*>(orow + d0) = pk;
          }
        }
    }
  }
  if constexpr (!PIPE) break;
  if (next >= a.BH) break;
  if (pre) load_kv(item_bh(next));     // a wave without a query block of its own
  item = next;
  __syncthreads();                     // every wave is done with this item's K / V image
  }
#endif
}

template <int DP, bool NR = false>
static int launch_attention(const AttnArgs& a, int batch, hipStream_t stream) {
  constexpr int lds = ATT_KEYS * (DP == 64 ? 128 : 256) + ATT_KEYS * (DP == 64 ? 128 : 256);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_kernel<DP, NR>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done = true;
  }
  AttnArgs b_ = a;
  b_.BH = batch * a.H;
  static const int xcd = [] { const char* e = getenv("DMVAE_ATTN_XCD"); return !(e && e[0] == '0') ? 1 : 0; }();
  b_.xcd = xcd;
  if constexpr (DP == 96 && !NR) {   // two or more (batch, head) items per CU: the persistent form that loads the next item's K / V under the current one's sweeps (DMVAE_ATTN_PIPE=0: off)
    static const int pipe = [] { const char* e = getenv("DMVAE_ATTN_PIPE"); return !(e && e[0] == '0') ? 1 : 0; }();
    static const int cus = [] {
      int dev = 0, n = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
      return n & ~7;
    }();
    if (pipe && b_.BH >= 2 * cus) {
      static bool attr_pipe = false;
      if (!attr_pipe) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_kernel<DP, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_pipe = true;
      }
      hipLaunchKernelGGL((attention_kernel<DP, false, true>), dim3(cus), dim3(ATTN_THREADS), lds, stream, b_);
      DMVAE_CHECK_LAUNCH();
      return 0;
    }
  }
  hipLaunchKernelGGL((attention_kernel<DP, NR>), dim3(batch * a.H), dim3(ATTN_THREADS), lds, stream, b_);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

}  // namespace dmvae_vit

extern "C" int dmvae_attention_qkv_lse_bf16(const void* qkv, void* out, void* lse, int batch, int seq, int heads, int head_dim, float scale, hipStream_t stream);
extern "C" int dmvae_attention_qkv_bf16(const void* qkv, void* out, int batch, int seq, int heads, int head_dim, float scale,
                                        hipStream_t stream) {
  return dmvae_attention_qkv_lse_bf16(qkv, out, nullptr, batch, seq, heads, head_dim, scale, stream);
}
extern "C" int dmvae_attention_qkv_lse_bf16(const void* qkv, void* out, void* lse, int batch, int seq, int heads, int head_dim, float scale, hipStream_t stream) {
  using namespace dmvae_vit;
  DMVAE_CHECK_ARG(qkv && out && batch > 0 && heads > 0 && seq > 0, "attention_qkv_bf16: bad argument");
  DMVAE_CHECK_ARG(head_dim == ATT_D && seq <= ATT_KEYS, "attention_qkv_bf16: needs head_dim 64 and seq <= 288 (got %d, %d)", head_dim, seq);
  const long long C = (long long)heads * head_dim;
  AttnArgs a = {};
  a.q = (const bf16*)qkv; a.k = a.q + C; a.v = a.q + 2 * C; a.out = (bf16*)out;
  a.q_bs = a.k_bs = a.v_bs = (long long)seq * 3 * C; a.q_hs = a.k_hs = a.v_hs = head_dim;
  a.q_rs = a.k_rs = a.v_rs = (int)(3 * C);
  a.S = seq; a.H = heads; a.D = head_dim; a.QD = head_dim; a.scale = scale; a.lse = (float*)lse;
  return launch_attention<64>(a, batch, stream);
}

// Same kernel on head-major operands (q, k: [B*H][S][Dp], v: [B*H][S][D]; LightningDiT after QK-norm + RoPE, head dim 64 or 72 -> Dp 64 / 96).
extern "C" int dmvae_attention_heads_lse_bf16(const void* q, const void* k, const void* v, void* out, void* lse, int batch, int seq, int heads, int head_dim,
                                              int head_dim_padded, float scale, hipStream_t stream);
extern "C" int dmvae_attention_heads_bf16(const void* q, const void* k, const void* v, void* out, int batch, int seq, int heads, int head_dim,
                                          int head_dim_padded, float scale, hipStream_t stream) {
  return dmvae_attention_heads_lse_bf16(q, k, v, out, nullptr, batch, seq, heads, head_dim, head_dim_padded, scale, stream);
}
extern "C" int dmvae_attention_heads_lse_bf16(const void* q, const void* k, const void* v, void* out, void* lse, int batch, int seq, int heads, int head_dim,
                                              int head_dim_padded, float scale, hipStream_t stream) {
  using namespace dmvae_vit;
  DMVAE_CHECK_ARG(q && k && v && out && batch > 0 && heads > 0 && seq > 0, "attention_heads_bf16: bad argument");
  // q / k rows: head_dim_padded channels -- 64 / 96 (zero-padded by the producer), or head_dim itself (no padding in memory; the kernels' 96-wide products see zeros)
  const int dpc = (head_dim_padded + 31) / 32 * 32;
  DMVAE_CHECK_ARG(seq <= ATT_KEYS && head_dim % 8 == 0 && head_dim <= head_dim_padded && (head_dim_padded == 64 || head_dim_padded == 96 || head_dim_padded == head_dim) &&
                  (dpc == 64 || dpc == 96),
                  "attention_heads_bf16: needs seq <= 288, head_dim %% 8 == 0, q / k rows of 64, 96 or head_dim <= 96 channels (got %d, %d, %d)", seq, head_dim, head_dim_padded);
  AttnArgs a = {};
  a.q = (const bf16*)q; a.k = (const bf16*)k; a.v = (const bf16*)v; a.out = (bf16*)out;
  a.q_hs = a.k_hs = (long long)seq * head_dim_padded; a.q_bs = a.k_bs = a.q_hs * heads;
  a.v_hs = (long long)seq * head_dim; a.v_bs = a.v_hs * heads;
  a.q_rs = a.k_rs = head_dim_padded; a.v_rs = head_dim;
  a.S = seq; a.H = heads; a.D = head_dim; a.QD = head_dim_padded; a.scale = scale; a.lse = (float*)lse;
  return dpc == 64 ? launch_attention<64>(a, batch, stream) : launch_attention<96>(a, batch, stream);
}

// LightningDiT's attention straight from the qkv Linear's output [B][N][3][H][D]: QK RMSNorm + weight, 2-D RoPE (what dmvae_qknorm_rope_bf16 does) applied
// while K is staged / Q fragments are loaded, then the same fused softmax(q k^T) v -- no head-major q / k / v round trip through HBM.
extern "C" int dmvae_attention_qknorm_rope_bf16(const void* qkv, const void* q_weight, const void* k_weight, const void* cos_table, const void* sin_table,
                                                void* out, int batch, int seq, int heads, int head_dim, float eps, float scale, hipStream_t stream) {
  using namespace dmvae_vit;
  DMVAE_CHECK_ARG(qkv && q_weight && k_weight && cos_table && sin_table && out && batch > 0 && heads > 0 && seq > 0, "attention_qknorm_rope_bf16: bad argument");
  DMVAE_CHECK_ARG(seq <= ATT_KEYS && head_dim % 8 == 0 && head_dim >= 8 && head_dim <= 96,
                  "attention_qknorm_rope_bf16: needs seq <= 288 and head_dim a multiple of 8 up to 96 (got %d, %d)", seq, head_dim);
  const long long C = (long long)heads * head_dim;
  AttnArgs a = {};
  a.q = (const bf16*)qkv; a.k = a.q + C; a.v = a.q + 2 * C; a.out = (bf16*)out;
  a.q_bs = a.k_bs = a.v_bs = (long long)seq * 3 * C; a.q_hs = a.k_hs = a.v_hs = head_dim;
  a.q_rs = a.k_rs = a.v_rs = (int)(3 * C);
  a.S = seq; a.H = heads; a.D = head_dim; a.QD = head_dim; a.scale = scale;
  a.qw = (const float*)q_weight; a.kw = (const float*)k_weight; a.cosb = (const float*)cos_table; a.sinb = (const float*)sin_table; a.eps = eps;
  return head_dim <= 64 ? launch_attention<64, true>(a, batch, stream) : launch_attention<96, true>(a, batch, stream);
}
