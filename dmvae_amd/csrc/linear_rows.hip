// nn.Linear on a HANDFUL of rows -- one row per SAMPLE -- under autocast(bf16): Y[M][N] = act(X[M][K] . W[N][K]^T + bias[N]), M <= 64.
//
// Replaces the library GEMV/GEMM behind the per-sample conditioning Linears of LightningDiT (diffusion/lightningdit/lightningdit.py):
//   adaLN_modulation = Sequential(SiLU, Linear(C, 6C)) of every block (:236-240) and Linear(C, 2C) of the final layer (:266-268),
//   TimestepEmbedder.mlp = Linear(256, C) -> SiLU -> Linear(C, C) (:96-139),
// and, against a transposed copy of the weight, their input gradients.  B = 16 ... 64 rows against a 6912 x 1152 weight: the weight is read once and nothing
// is reused -- bound by streaming W from HBM (15.9 MB per adaLN call = 3.2 us at 5 TB/s), which neither tile kernel of this build is shaped for (the 256-row
// GEMM tile is 94 % padding; the small batched kernel measured 62 us per call against ~5 for the vendor library, DESIGN_HISTORY.md 9.6).
//
// One workgroup (4 waves) owns 16 output columns; its waves split the K steps round-robin (so that ~7 waves per CU keep ~60 KB of weight reads in flight with
// N / 16 workgroups over 256 CUs), every wave issues ALL of its W loads up front, multiplies on the matrix cores (v_mfma_f32_16x16x32_bf16: W rows as the A operand
// straight from global memory -- lane l reads 16 B of row l & 15 at K offset 8 (l >> 4), the four lanes of a row covering 64 contiguous bytes -- X^T as the B operand
// from L1/L2: every workgroup re-reads the same <= 147 KB of X), and the four partial accumulators are summed in wave order through LDS: deterministic, no atomics.
// The activation (SiLU) is applied to the bf16-ROUNDED pre-activation: bit-identical to this call with act = 0 followed by dmvae_silu_fwd (Linear -> SiLU under autocast).
#include "common.h"
#include "dmvae_hip.h"

namespace dmvae_linear_rows {

constexpr int MAXSTEPS = 16;  // K steps per wave and pass (all of their weight loads in flight together); longer reductions loop

struct Args {
  const bf16* x; const bf16* w; const void* bias; void* y;
  int M, N, K, ldx, ldw, ldy, act, bias_bf16, out_f32, w_layout;
  // batched form (dmvae_linear_rows_batched_bf16: blockIdx.y = layer; the adaLN Linears of every block in one launch): per-layer weight / bias pointers,
  // element strides between the layers' x and y (xs = 0: one x for all layers)
  const void* const* wtab; const void* const* btab; long long xs, ys;
};

// G = row groups of 16 (M <= 64); WAVES = waves of the workgroup that share the K steps: 4 when N / 16 workgroups already fill the chip several times over
// (adaLN: 432), 16 (8 from M > 32 on: LDS) when few columns meet a long reduction (the input gradient of adaLN: 72 workgroups x 216 K steps)
template <int G, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void linear_rows_kernel(Args a) {
  static_assert(G <= WAVES, "one finishing wave per row group");
  __shared__ float red[WAVES][G][64][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = blockIdx.x * 16;
  if (a.wtab) {   // layer blockIdx.y of the batched form (block-uniform)
    const int l = blockIdx.y;
    a.w = (const bf16*)a.wtab[l];
    a.bias = a.btab ? a.btab[l] : nullptr;
    a.x += (size_t)l * a.xs;
    a.y = a.out_f32 ? (void*)((float*)a.y + (size_t)l * a.ys) : (void*)((bf16*)a.y + (size_t)l * a.ys);
  }
  const int r = lane & 15, kc = (lane >> 4) * 8;
  const int nrow = n0 + r < a.N ? n0 + r : a.N - 1;            // rows past N (N % 16 != 0): read a valid row, never stored
  // w_layout 1: K-tile-major [K / 32][N][32] (dmvae_linear_weight_t_kmajor / dmvae_pack_conv_weight_v2's second copy): the 16 rows of a K step are one
  // contiguous KiB; 0: row-major [N][ldw]
  const bf16* wp = a.w_layout ? a.w + (size_t)nrow * 32 + kc : a.w + (size_t)nrow * a.ldw + kc;
  const size_t wstep = a.w_layout ? (size_t)a.N * 32 : 32;
  const bf16* xp[G];
#pragma unroll
  for (int g = 0; g < G; g++) {
    const int m = g * 16 + r;
    xp[g] = a.x + (size_t)(m < a.M ? m : 0) * a.ldx + kc;       // rows past M: a valid row, never stored
  }
  f32x4 acc[G];
#pragma unroll
  for (int g = 0; g < G; g++) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int ksteps = a.K >> 5;
  for (int s0 = wave; s0 < ksteps; s0 += WAVES * MAXSTEPS) {
    bf16x8 wf[MAXSTEPS];
#pragma unroll
    for (int i = 0; i < MAXSTEPS; i++) {                       // every weight load of this pass in flight before the first use
      const int s = s0 + i * WAVES;
      if (s < ksteps) wf[i] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp + (size_t)s * wstep));
    }
#pragma unroll
    for (int i = 0; i < MAXSTEPS; i++) {
      const int s = s0 + i * WAVES;
      if (s < ksteps) {
#pragma unroll
        for (int g = 0; g < G; g++) {
          const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xp[g] + (size_t)s * 32);
          acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf, acc[g], 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int g = 0; g < G; g++)
#pragma unroll
    for (int j = 0; j < 4; j++) red[wave][g][lane][j] = acc[g][j];
  __syncthreads();
  // wave w finishes row group w (G <= WAVES): lane l holds sample m = 16 g + (l & 15), columns n0 + 4 (l >> 4) + j
  if (wave < G) {
    const int g = wave;
    const int m = g * 16 + r, nb = n0 + (lane >> 4) * 4;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      float s = red[0][g][lane][j];
#pragma unroll
      for (int w = 1; w < WAVES; w++) s += red[w][g][lane][j];   // fixed order
      if (a.bias && nb + j < a.N) s += a.bias_bf16 ? (float)((const bf16*)a.bias)[nb + j] : ((const float*)a.bias)[nb + j];
      v[j] = s;
    }
    if (m < a.M) {
      if (a.out_f32) {
        float* o = (float*)a.y + (size_t)m * a.ldy + nb;
#pragma unroll
        for (int j = 0; j < 4; j++)
          if (nb + j < a.N) o[j] = v[j];
      } else {
        bf16* o = (bf16*)a.y + (size_t)m * a.ldy + nb;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          bf16 t = (bf16)v[j];
          if (a.act == 1) { const float f = (float)t; t = (bf16)(f * sigmoidf_(f)); }      // dmvae_silu_fwd's arithmetic on the rounded pre-activation
          if (nb + j < a.N) o[j] = t;
        }
      }
    }
  }
}

// Weight / bias gradient of the same Linear: dW[N][K] (+)= dY[M][N]^T . X[M][K], db[N] (+)= sum_m dY[m][n], M <= 64 -- an outer-product accumulation with a
// reduction of at most 64 terms per element: nothing to tile, bound by WRITING the f32 gradient (32 MB for adaLN's 6912 x 1152).  One workgroup = 16 rows of dW
// x a 1024-column chunk: X's chunk and dY's 16 columns staged in LDS, a thread owns four consecutive columns of all 16 rows (64 f32 accumulators), summed over the
// samples in order; float4 stores, a row's 4 KiB contiguous.
__global__ __launch_bounds__(256) void linear_rows_wgrad_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x, float* __restrict__ dw, float* __restrict__ db,
                                                                int M, int N, int K, int lddy, int ldx, int accumulate) {
  extern __shared__ char smem[];
  bf16* xs = reinterpret_cast<bf16*>(smem);                   // [M][1024]
  float* ds = reinterpret_cast<float*>(smem + (size_t)M * 2048);   // [M][16]
  const int n0 = blockIdx.x * 16, k0 = blockIdx.y * 1024, t = threadIdx.x;
  const int kw = min(1024, K - k0);                           // columns of this chunk (a multiple of 8)
  for (int i = t; i < M * 128; i += 256) {                    // 16-B pieces of X's chunk
    const int m = i >> 7, c = (i & 127) * 8;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (c < kw) v = *reinterpret_cast<const uint4*>(x + (size_t)m * ldx + k0 + c);
    *reinterpret_cast<uint4*>(xs + m * 1024 + c) = v;
  }
  for (int i = t; i < M * 16; i += 256) {
    const int m = i >> 4, j = i & 15;
    ds[i] = n0 + j < N ? (float)dy[(size_t)m * lddy + n0 + j] : 0.f;
  }
  __syncthreads();
  if (db && blockIdx.y == 0 && t < 16 && n0 + t < N) {
    float s = 0.f;
    for (int m = 0; m < M; m++) s += ds[m * 16 + t];
    db[n0 + t] = (accumulate ? db[n0 + t] : 0.f) + s;
  }
  const int c = t * 4;
  if (c >= kw) return;
  float acc[16][4];
#pragma unroll
  for (int j = 0; j < 16; j++)
#pragma unroll
    for (int e = 0; e < 4; e++) acc[j][e] = 0.f;
  for (int m = 0; m < M; m++) {
    const bf16x4 xv = *reinterpret_cast<const bf16x4*>(xs + m * 1024 + c);
    const float x0 = (float)xv[0], x1 = (float)xv[1], x2 = (float)xv[2], x3 = (float)xv[3];
    const float4* dp = reinterpret_cast<const float4*>(ds + m * 16);      // the same address for every lane: LDS broadcast
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const float4 d = dp[q];
      const float dv[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int j = q * 4 + e;
        acc[j][0] += dv[e] * x0; acc[j][1] += dv[e] * x1; acc[j][2] += dv[e] * x2; acc[j][3] += dv[e] * x3;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 16; j++) {
    if (n0 + j >= N) break;
    float4* o = reinterpret_cast<float4*>(dw + (size_t)(n0 + j) * K + k0 + c);
    float4 v = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
    if (accumulate) { const float4 p = *o; v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w; }
    __builtin_nontemporal_store(v.x, &o->x); __builtin_nontemporal_store(v.y, &o->y); __builtin_nontemporal_store(v.z, &o->z); __builtin_nontemporal_store(v.w, &o->w);
  }
}

}  // namespace dmvae_linear_rows

extern "C" int dmvae_linear_rows_supported(int M, int N, int K) { return M >= 1 && M <= 64 && N >= 4 && N % 4 == 0 && K >= 32 && K % 32 == 0; }

extern "C" int dmvae_linear_rows_bf16(const void* x, const void* w, const void* bias, void* y, int M, int N, int K, int ldx, int ldw, int ldy, int act,
                                      int bias_bf16, int out_f32, int w_layout, hipStream_t stream) {
  using namespace dmvae_linear_rows;
  DMVAE_CHECK_ARG(x && w && y, "linear_rows_bf16: null pointer");
  DMVAE_CHECK_ARG(dmvae_linear_rows_supported(M, N, K), "linear_rows_bf16: M=%d N=%d K=%d (1 <= M <= 64, N %% 4 == 0, K %% 32 == 0)", M, N, K);
  DMVAE_CHECK_ARG(ldx >= K && (w_layout == 1 || (ldw >= K && ldw % 8 == 0)) && ldy >= N && ldx % 8 == 0 && ldy % 4 == 0, "linear_rows_bf16: leading dimensions ldx=%d ldw=%d ldy=%d", ldx, ldw, ldy);
  DMVAE_CHECK_ARG(w_layout == 0 || w_layout == 1, "linear_rows_bf16: w_layout must be 0 (row-major) or 1 (K-tile-major)");
  DMVAE_CHECK_ARG(act == 0 || (act == 1 && !out_f32), "linear_rows_bf16: act must be 0 (none) or 1 (SiLU, bf16 result)");
  DMVAE_CHECK_ARG(((uintptr_t)x | (uintptr_t)w) % 16 == 0 && (uintptr_t)y % 8 == 0, "linear_rows_bf16: operands must be 16-byte aligned");
  Args a{(const bf16*)x, (const bf16*)w, bias, y, M, N, K, ldx, ldw, ldy, act, bias_bf16, out_f32, w_layout, nullptr, nullptr, 0, 0};
  const dim3 grid((N + 15) / 16);
  const int G = (M + 15) / 16;
  const bool deep = (N + 15) / 16 < 256 && K >= 1024;      // few workgroups, long reduction: more waves per workgroup share the K steps
#define DMVAE_LR(GG, WW) hipLaunchKernelGGL((linear_rows_kernel<GG, WW>), grid, dim3(WW * 64), 0, stream, a)
  switch (G) {
    case 1: if (deep) DMVAE_LR(1, 16); else DMVAE_LR(1, 4); break;
    case 2: if (deep) DMVAE_LR(2, 16); else DMVAE_LR(2, 4); break;
    case 3: if (deep) DMVAE_LR(3, 8); else DMVAE_LR(3, 4); break;
    default: if (deep) DMVAE_LR(4, 8); else DMVAE_LR(4, 4); break;
  }
#undef DMVAE_LR
  DMVAE_CHECK_LAUNCH();
  return 0;
}

extern "C" int dmvae_linear_rows_batched_bf16(const void* x, long long x_layer_stride, const void* w_table, const void* bias_table, void* y, long long y_layer_stride,
                                              int layers, int M, int N, int K, int ldx, int ldw, int ldy, int act, int bias_bf16, int out_f32, int w_layout,
                                              hipStream_t stream) {
  using namespace dmvae_linear_rows;
  DMVAE_CHECK_ARG(x && w_table && y && layers > 0 && layers <= 65535, "linear_rows_batched_bf16: null pointer or bad layer count");
  DMVAE_CHECK_ARG(dmvae_linear_rows_supported(M, N, K), "linear_rows_batched_bf16: M=%d N=%d K=%d (1 <= M <= 64, N %% 4 == 0, K %% 32 == 0)", M, N, K);
  DMVAE_CHECK_ARG(ldx >= K && (w_layout == 1 || (ldw >= K && ldw % 8 == 0)) && ldy >= N && ldx % 8 == 0 && ldy % 4 == 0 && x_layer_stride % 8 == 0 && y_layer_stride % 4 == 0,
                  "linear_rows_batched_bf16: leading dimensions ldx=%d ldw=%d ldy=%d", ldx, ldw, ldy);
  DMVAE_CHECK_ARG(w_layout == 0 || w_layout == 1, "linear_rows_batched_bf16: w_layout must be 0 (row-major) or 1 (K-tile-major)");
  DMVAE_CHECK_ARG(act == 0 || (act == 1 && !out_f32), "linear_rows_batched_bf16: act must be 0 (none) or 1 (SiLU, bf16 result)");
  DMVAE_CHECK_ARG((uintptr_t)x % 16 == 0 && (uintptr_t)y % 8 == 0, "linear_rows_batched_bf16: operands must be 16-byte aligned");
  Args a{(const bf16*)x, nullptr, nullptr, y, M, N, K, ldx, ldw, ldy, act, bias_bf16, out_f32, w_layout, (const void* const*)w_table, (const void* const*)bias_table,
         x_layer_stride, y_layer_stride};
  const dim3 grid((N + 15) / 16, layers);
  const int G = (M + 15) / 16;
  const bool deep = (long long)((N + 15) / 16) * layers < 256 && K >= 1024;
#define DMVAE_LR(GG, WW) hipLaunchKernelGGL((linear_rows_kernel<GG, WW>), grid, dim3(WW * 64), 0, stream, a)
  switch (G) {
    case 1: if (deep) DMVAE_LR(1, 16); else DMVAE_LR(1, 4); break;
    case 2: if (deep) DMVAE_LR(2, 16); else DMVAE_LR(2, 4); break;
    case 3: if (deep) DMVAE_LR(3, 8); else DMVAE_LR(3, 4); break;
    default: if (deep) DMVAE_LR(4, 8); else DMVAE_LR(4, 4); break;
  }
#undef DMVAE_LR
  DMVAE_CHECK_LAUNCH();
  return 0;
}

extern "C" int dmvae_linear_rows_wgrad(const void* dy, const void* x, void* dw, void* db, int M, int N, int K, int lddy, int ldx, int accumulate, hipStream_t stream) {
  using namespace dmvae_linear_rows;
  DMVAE_CHECK_ARG(dy && x && dw, "linear_rows_wgrad: null pointer");
  DMVAE_CHECK_ARG(M >= 1 && M <= 64 && N >= 1 && K >= 8 && K % 8 == 0, "linear_rows_wgrad: M=%d N=%d K=%d (1 <= M <= 64, K %% 8 == 0)", M, N, K);
  DMVAE_CHECK_ARG(lddy >= N && ldx >= K && ldx % 8 == 0, "linear_rows_wgrad: leading dimensions lddy=%d ldx=%d", lddy, ldx);
  DMVAE_CHECK_ARG((uintptr_t)x % 16 == 0 && (uintptr_t)dw % 16 == 0, "linear_rows_wgrad: x and dw must be 16-byte aligned");
  const size_t lds = (size_t)M * 2048 + (size_t)M * 64;
  static bool attr_done = false;
  if (!attr_done) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(linear_rows_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 2048 + 64 * 64); attr_done = true; }
  hipLaunchKernelGGL(linear_rows_wgrad_kernel, dim3((N + 15) / 16, (K + 1023) / 1024), dim3(256), lds, stream, (const bf16*)dy, (const bf16*)x, (float*)dw, (float*)db, M, N, K, lddy, ldx,
                     accumulate);
  DMVAE_CHECK_LAUNCH();
  return 0;
}
