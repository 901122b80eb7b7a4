// Whole-stack backward pieces of LightningDiT's training route (diffusion/lightningdit/lightningdit.py:236-250 x depth; the student's flow-matching step,
// train_dmd.py:565-575, train_diffusion.py:290-297) -- what functional.DitStackFn adds to csrc/dit.hip's per-block kernels:
//
//   rms_gate_bwd_kernel<APPLY, GATE>   one pass over the f32 residual-stream gradient dt per sub-layer boundary:
//        APPLY: dt += RMSNorm+modulate backward of da (the row statistics come from dit.hip::rmsnorm_rowstat_kernel), with the per-sample partial sums of
//               d shift / d scale / d norm-weight;
//        GATE:  dy = bf16(gate[b] * dt) of the NEXT gated residual down the backward pass (its branch output y read here), with the partial sums of d gate.
//        The per-block route ran these as two kernels (rmsnorm_modulate_bwd_apply_kernel + gated_residual_bwd_kernel, the second on 80 workgroups); fused, dt is
//        read once and written once per boundary.
//   dit_bwd_finalize / dit_bwd_weight  ONE reduction of every boundary's partial sums at the end of the backward pass (57 boundaries for 28 blocks) into the
//        bf16 adaLN-chunk gradients d mod [L][B][6C] and the 2 L norm-weight gradients, instead of two small launches per boundary.
//   rows_wgrad_mfma_kernel             weight + bias gradient of L per-sample Linears (adaLN_modulation[1] of every block) in one launch on the matrix cores:
//        dW_l [N][K] = dY_l [M][N]^T . X [M][K], M <= 64 samples.
//   colsum2_batched_kernel             the deferred second stage of qknorm_rope_bwd's norm-weight gradients, all layers in one launch.
// Every reduction is two-stage with a fixed order: reruns are bit-identical.
#include "common.h"
#include "dmvae_hip.h"

namespace dmvae_dit_stack {

// grid (bps, B); thread t owns columns 4 t .. 4 t + 3 of every row of its block's row range (fully coalesced row accesses, no cross-lane work).
// part: [B][bps][4][C]: d shift | d scale | d norm weight | d gate contributions of the block (slots of a disabled half are not written).
template <bool APPLY, bool GATE>
__global__ __launch_bounds__(512) void rms_gate_bwd_kernel(const bf16* __restrict__ da, const float* __restrict__ x, const float* __restrict__ w,
                                                           const bf16* __restrict__ mod, const float2* __restrict__ rowstat, float* __restrict__ dx_io,
                                                           float* __restrict__ part, int N, int C, int stride, int scale_off, const bf16* __restrict__ y,
                                                           const bf16* __restrict__ gmod, int gstride, int gate_off, bf16* __restrict__ dy) {
  const int b = blockIdx.y, c = threadIdx.x * 4;
  if (c >= C) return;
  f32x4 gw = {0, 0, 0, 0}, gm = {0, 0, 0, 0}, gg = {0, 0, 0, 0};
  f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0}, a3 = {0, 0, 0, 0};
  if constexpr (APPLY) {
    gw = *reinterpret_cast<const f32x4*>(w + c);
    const bf16x4 sc = *reinterpret_cast<const bf16x4*>(mod + (size_t)b * stride + scale_off + c);
#pragma unroll
    for (int e = 0; e < 4; e++) gm[e] = (float)(bf16)(1.f + (float)sc[e]);   // `1 + scale` is a bf16 tensor in the reference's autocast graph
  }
  if constexpr (GATE) {
    const bf16x4 g = *reinterpret_cast<const bf16x4*>(gmod + (size_t)b * gstride + gate_off + c);
#pragma unroll
    for (int e = 0; e < 4; e++) gg[e] = (float)g[e];
  }
  const int rpb = (N + gridDim.x - 1) / gridDim.x;
  const int n1 = min(N, (int)(blockIdx.x + 1) * rpb);
#pragma unroll 2
  for (int n = blockIdx.x * rpb; n < n1; n++) {
    const size_t off = ((size_t)b * N + n) * C + c;
    f32x4 o = *reinterpret_cast<const f32x4*>(dx_io + off);
    if constexpr (APPLY) {
      const float2 st = rowstat[(size_t)b * N + n];
      const f32x4 v = *reinterpret_cast<const f32x4*>(x + off);
      const bf16x4 d = *reinterpret_cast<const bf16x4*>(da + off);
#pragma unroll
      for (int e = 0; e < 4; e++) {   // dit.hip::rmsnorm_modulate_bwd_apply_kernel's expressions
        const float nh = v[e] * st.x, dv = (float)d[e];
        const float g = dv * gw[e] * gm[e];
        o[e] += st.x * (g - nh * st.y);
        a0[e] += dv; a1[e] += dv * nh * gw[e]; a2[e] += dv * gm[e] * nh;
      }
      *reinterpret_cast<f32x4*>(dx_io + off) = o;
    }
    if constexpr (GATE) {   // dit.hip::gated_residual_bwd_kernel's expressions
      const bf16x4 yv = *reinterpret_cast<const bf16x4*>(y + off);
      bf16x4 q;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        q[e] = (bf16)(gg[e] * o[e]);
        a3[e] += o[e] * (float)yv[e];
      }
      *reinterpret_cast<bf16x4*>(dy + off) = q;
    }
  }
  float* po = part + ((size_t)b * gridDim.x + blockIdx.x) * 4 * C;
  if constexpr (APPLY) {
    *reinterpret_cast<f32x4*>(po + c) = a0;
    *reinterpret_cast<f32x4*>(po + C + c) = a1;
    *reinterpret_cast<f32x4*>(po + 2 * C + c) = a2;
  }
  if constexpr (GATE) *reinterpret_cast<f32x4*>(po + 3 * C + c) = a3;
}

// rstd and m2 = mean_c(g * xhat), g = da * w * bf16(1 + scale), per row -- one wave per row (dit.hip::rmsnorm_rowstat_kernel, restated here: kernels are per translation unit)
__global__ __launch_bounds__(256) void rowstat_kernel(const bf16* __restrict__ da, const float* __restrict__ x, const float* __restrict__ w, const bf16* __restrict__ mod,
                                                      float2* __restrict__ rowstat, int rows, int N, int C, int stride, int scale_off, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const bf16* mrow = mod + (size_t)(row / N) * stride + scale_off;
  float ss = 0.f, s2 = 0.f;
  for (int c = lane * 4; c < C; c += 256) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)row * C + c);
    const bf16x4 d = *reinterpret_cast<const bf16x4*>(da + (size_t)row * C + c);
    const f32x4 gw = *reinterpret_cast<const f32x4*>(w + c);
    const bf16x4 sc = *reinterpret_cast<const bf16x4*>(mrow + c);
#pragma unroll
    for (int e = 0; e < 4; e++) {
      ss += v[e] * v[e];
      s2 += (float)d[e] * gw[e] * (float)(bf16)(1.f + (float)sc[e]) * v[e];
    }
  }
  const float rs = rsqrtf(wave_sum(ss) / (float)C + eps);
  const float m2 = wave_sum(s2) * rs / (float)C;
  if (lane == 0) rowstat[row] = make_float2(rs, m2);
}

// One reduction for the whole stack.  Boundary slots (dmvae_dit_stack_slots): s = 2 l: norm1 of block l (+ the MLP gate of block l - 1 when l > 0);
// s = 2 l + 1: norm2 of block l + the attention gate of block l; s = 2 L: the MLP gate of block L - 1 alone.  part: [slots][B][bps][4][C].
// dmod: bf16 [L][B][6 C] in adaLN chunk order (shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp); wpart: f32 [2 L][B][C].
__global__ __launch_bounds__(256) void dit_bwd_finalize_kernel(const float* __restrict__ part, bf16* __restrict__ dmod, float* __restrict__ wpart, int L, int B, int bps,
                                                               int C) {
  const int c = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y, s = blockIdx.z;
  if (c >= C) return;
  const bool has_apply = s < 2 * L, has_gate = s > 0;
  const float* q = part + (((size_t)s * B + b) * bps) * 4 * C + c;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  for (int k = 0; k < bps; k++, q += 4 * (size_t)C) {
    if (has_apply) { s0 += q[0]; s1 += q[C]; s2 += q[2 * (size_t)C]; }
    if (has_gate) s3 += q[3 * (size_t)C];
  }
  const int l = s >> 1, odd = s & 1;
  const size_t row = 6 * (size_t)C;
  if (has_apply) {
    bf16* m = dmod + ((size_t)l * B + b) * row;
    m[(odd ? 3 : 0) * (size_t)C + c] = (bf16)s0;
    m[(odd ? 4 : 1) * (size_t)C + c] = (bf16)s1;
    wpart[((size_t)s * B + b) * C + c] = s2;
  }
  if (has_gate) {
    const int lg = odd ? l : l - 1;   // s = 2 L: l = L, even: the last block's MLP gate
    dmod[((size_t)lg * B + b) * row + (odd ? 2 : 5) * (size_t)C + c] = (bf16)s3;
  }
}
// dw_tab[s] (f32 [C], the flat-buffer gradient slice of block s / 2's norm1 (even s) or norm2 (odd s) weight) (+)= sum_b wpart[s][b][c]
__global__ __launch_bounds__(256) void dit_bwd_weight_kernel(const float* __restrict__ wpart, float* const* __restrict__ dw_tab, int B, int C, int accumulate) {
  const int c = blockIdx.x * 256 + threadIdx.x, s = blockIdx.y;
  if (c >= C) return;
  float a = 0.f;
  for (int b = 0; b < B; b++) a += wpart[((size_t)s * B + b) * C + c];
  float* dw = dw_tab[s];
  dw[c] = (accumulate ? dw[c] : 0.f) + a;
}

// part [L][nblk][2][D] -> o0_tab[l][d], o1_tab[l][d]: dit.hip::colsum2_kernel for every layer at once, 64 row groups per column (fixed assignment, fixed-order combine)
__global__ __launch_bounds__(1024) void colsum2_batched_kernel(const float* __restrict__ part, float* const* __restrict__ o0_tab, float* const* __restrict__ o1_tab,
                                                               int nblk, int D, int accumulate) {
  __shared__ float red[64][17];
  const int col = threadIdx.x & 15, grp = threadIdx.x >> 4, l = blockIdx.y;
  const int i = blockIdx.x * 16 + col;
  const float* p = part + (size_t)l * nblk * 2 * D;
  float a = 0.f;
  if (i < 2 * D) {
#pragma unroll 8
    for (int b = grp; b < nblk; b += 64) a += p[(size_t)b * 2 * D + i];
  }
  red[grp][col] = a;
  __syncthreads();
  if (threadIdx.x < 16 && i < 2 * D) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 64; g++) t += red[g][col];
    const int which = i / D, d = i - which * D;
    float* o = which ? o1_tab[l] : o0_tab[l];
    o[d] = (accumulate ? o[d] : 0.f) + t;
  }
}

// dW_l [N][K] f32 = dY_l [M][N]^T . X [M][K] (M <= 64; x given TRANSPOSED: xT [K][MP] bf16, MP = 32 or 64, zero beyond M), db_l [N] = column sums of dY_l.
// grid (N / 16, ceil(K / 512), L); a wave owns 128 columns of K as 8 MFMA column blocks whose lane-column c maps to k = k0 + 8 c + t: a lane ends up with 8
// CONSECUTIVE k of each of its 4 rows -> two 16-B stores per row, the 16 lanes of a group 512 contiguous bytes.  dY's 16 columns go through LDS transposed
// (the reduction dim m is the slow one in dY); v_mfma_f32_16x16x32_bf16: A = dY^T (row n, reduction m), B = X (reduction m, column k).
template <int MP>
__global__ __launch_bounds__(256) void rows_wgrad_mfma_kernel(const bf16* __restrict__ dy, long long dys, const bf16* __restrict__ xT, float* const* __restrict__ dw_tab,
                                                              float* const* __restrict__ db_tab, float* dw_one, float* db_one, int M, int N, int K, int lddy,
                                                              int accumulate) {
  __shared__ __attribute__((aligned(16))) bf16 dyt[16][MP + 8];   // +8: 16-B aligned rows on distinct banks
  const int l = blockIdx.z, n0 = blockIdx.x * 16, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const bf16* dyl = dy + (size_t)l * dys;
  for (int i = t; i < MP * 16; i += 256) {
    const int m = i >> 4, j = i & 15;
    dyt[j][m] = (m < M && n0 + j < N) ? dyl[(size_t)m * lddy + n0 + j] : (bf16)0.f;
  }
  __syncthreads();
  const int c = lane & 15, g = lane >> 4;
  const int k0 = (blockIdx.y * 4 + wave) * 128;
  float* dw = dw_tab ? dw_tab[l] : dw_one;
  float* db = dw_tab ? (db_tab ? db_tab[l] : nullptr) : db_one;
  bf16x8 af[MP / 32];
#pragma unroll
  for (int s = 0; s < MP / 32; s++) af[s] = *reinterpret_cast<const bf16x8*>(&dyt[c][8 * g + 32 * s]);
  if (db && blockIdx.y == 0 && wave == 0) {   // db[n0 + c] = sum_m dY[m][n0 + c]: the lane's 8 (16) values, then the four lane groups in a fixed order
    float s8 = 0.f;
#pragma unroll
    for (int s = 0; s < MP / 32; s++)
#pragma unroll
      for (int e = 0; e < 8; e++) s8 += (float)af[s][e];
    s8 += __shfl_xor(s8, 16, 64);
    s8 += __shfl_xor(s8, 32, 64);
    if (g == 0 && n0 + c < N) db[n0 + c] = (accumulate ? db[n0 + c] : 0.f) + s8;
  }
  if (k0 >= K) return;
  f32x4 acc[8];
#pragma unroll
  for (int u = 0; u < 8; u++) {
    acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int k = k0 + 8 * c + u;
    const bf16* xr = xT + (size_t)(k < K ? k : 0) * MP + 8 * g;
#pragma unroll
    for (int s = 0; s < MP / 32; s++) {
      const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xr + 32 * s);
      acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[s], xf, acc[u], 0, 0, 0);
    }
  }
  // acc[u][r] = dW[n0 + 4 g + r][k0 + 8 c + u]
  if (k0 + 8 * c >= K) return;   // K % 8 == 0: a lane's 8 columns are all inside or all outside
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int n = n0 + 4 * g + r;
    if (n >= N) break;
    float4* o = reinterpret_cast<float4*>(dw + (size_t)n * K + k0 + 8 * c);
    float4 v0 = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]), v1 = make_float4(acc[4][r], acc[5][r], acc[6][r], acc[7][r]);
    if (accumulate) {
      const float4 p0 = o[0], p1 = o[1];
      v0.x += p0.x; v0.y += p0.y; v0.z += p0.z; v0.w += p0.w; v1.x += p1.x; v1.y += p1.y; v1.z += p1.z; v1.w += p1.w;
    }
    o[0] = v0; o[1] = v1;
  }
}

}  // namespace dmvae_dit_stack
using namespace dmvae_dit_stack;

static inline int stack_bps(int batch) {   // blocks per sample of the boundary kernels: ~1024 workgroups in all (dit.hip's rule)
  int bps = 1024 / batch;
  return bps > 32 ? 32 : (bps < 4 ? 4 : bps);
}
extern "C" int dmvae_dit_stack_bps(int batch) { return stack_bps(batch); }
extern "C" size_t dmvae_dit_stack_part_bytes(int layers, int batch, int c) {
  return (size_t)(2 * layers + 1) * batch * stack_bps(batch) * 4 * (size_t)c * sizeof(float);
}
extern "C" size_t dmvae_dit_stack_workspace(int layers, int batch, int seq, int c) {   // wpart [2 L][B][C] f32 + rowstat [B * seq] float2
  return (size_t)2 * layers * batch * (size_t)c * sizeof(float) + (size_t)batch * seq * sizeof(float2);
}

extern "C" int dmvae_dit_boundary_bwd(const void* da, const void* x, const void* w, const void* mod, int mod_stride, int scale_off, float eps, void* dx_io,
                                      const void* y, const void* gate_mod, int gate_stride, int gate_off, void* dy, void* part_slot, void* rowstat,
                                      int batch, int seq, int c, hipStream_t stream) {
  const bool apply = da != nullptr, gate = y != nullptr;
  DMVAE_CHECK_ARG((apply || gate) && dx_io && part_slot && batch > 0 && seq > 0, "dit_boundary_bwd: nothing to do or null buffers");
  DMVAE_CHECK_ARG(c % 4 == 0 && c >= 4 && c <= 2048, "dit_boundary_bwd: width must be a multiple of 4 up to 2048 (got %d)", c);
  DMVAE_CHECK_ARG(!apply || (x && w && mod && rowstat && scale_off >= 0 && scale_off % 4 == 0 && mod_stride % 4 == 0 && scale_off + c <= mod_stride),
                  "dit_boundary_bwd: the norm half needs x, w, mod, rowstat and a scale offset that is a multiple of 4 inside the modulation row");
  DMVAE_CHECK_ARG(!gate || (gate_mod && dy && gate_off >= 0 && gate_off % 4 == 0 && gate_stride % 4 == 0 && gate_off + c <= gate_stride),
                  "dit_boundary_bwd: the gate half needs gate_mod, dy and a gate offset that is a multiple of 4 inside the modulation row");
  const int rows = batch * seq, bps = stack_bps(batch);
  if (apply) {
    hipLaunchKernelGGL(rowstat_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, (const bf16*)da, (const float*)x, (const float*)w, (const bf16*)mod,
                       (float2*)rowstat, rows, seq, c, mod_stride, scale_off, eps);
    DMVAE_CHECK_LAUNCH();
  }
  const int bt = ((c / 4) + 63) / 64 * 64;
#define DMVAE_BND(A, G)                                                                                                                                        \
  hipLaunchKernelGGL((rms_gate_bwd_kernel<A, G>), dim3(bps, batch), dim3(bt), 0, stream, (const bf16*)da, (const float*)x, (const float*)w, (const bf16*)mod, \
                     (const float2*)rowstat, (float*)dx_io, (float*)part_slot, seq, c, mod_stride, scale_off, (const bf16*)y, (const bf16*)gate_mod, gate_stride,  \
                     gate_off, (bf16*)dy)
  if (apply && gate) DMVAE_BND(true, true);
  else if (apply) DMVAE_BND(true, false);
  else DMVAE_BND(false, true);
#undef DMVAE_BND
  DMVAE_CHECK_LAUNCH();
  return 0;
}

extern "C" int dmvae_dit_stack_finalize(const void* part, void* dmod, void* workspace, size_t workspace_bytes, const void* dw_table, int layers, int batch, int seq,
                                        int c, int accumulate, hipStream_t stream) {
  DMVAE_CHECK_ARG(part && dmod && workspace && dw_table && layers > 0 && batch > 0 && c > 0 && c % 4 == 0, "dit_stack_finalize: bad argument");
  DMVAE_CHECK_ARG(workspace_bytes >= dmvae_dit_stack_workspace(layers, batch, seq, c), "dit_stack_finalize: workspace too small");
  hipLaunchKernelGGL(dit_bwd_finalize_kernel, dim3((c + 255) / 256, batch, 2 * layers + 1), dim3(256), 0, stream, (const float*)part, (bf16*)dmod, (float*)workspace,
                     layers, batch, stack_bps(batch), c);
  DMVAE_CHECK_LAUNCH();
  hipLaunchKernelGGL(dit_bwd_weight_kernel, dim3((c + 255) / 256, 2 * layers), dim3(256), 0, stream, (const float*)workspace, (float* const*)dw_table, batch, c,
                     accumulate);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

extern "C" int dmvae_colsum2_batched(const void* part, const void* o0_table, const void* o1_table, int layers, int nblk, int d, int accumulate, hipStream_t stream) {
  DMVAE_CHECK_ARG(part && o0_table && o1_table && layers > 0 && nblk > 0 && d > 0, "colsum2_batched: bad argument");
  hipLaunchKernelGGL(colsum2_batched_kernel, dim3((2 * d + 15) / 16, layers), dim3(1024), 0, stream, (const float*)part, (float* const*)o0_table, (float* const*)o1_table,
                     nblk, d, accumulate);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

extern "C" int dmvae_linear_rows_wgrad_batched(const void* dy, long long dy_layer_stride, const void* xT, int mp, const void* dw_table, const void* db_table, void* dw,
                                               void* db, int layers, int M, int N, int K, int lddy, int accumulate, hipStream_t stream) {
  DMVAE_CHECK_ARG(dy && xT && (dw_table || dw) && layers > 0, "linear_rows_wgrad_batched: null pointer");
  DMVAE_CHECK_ARG(M >= 1 && M <= 64 && (mp == 32 || mp == 64) && mp >= M && N >= 1 && K >= 8 && K % 8 == 0 && lddy >= N,
                  "linear_rows_wgrad_batched: M=%d (<= 64, <= mp = %d in {32, 64}) N=%d K=%d (K %% 8 == 0) lddy=%d", M, mp, N, K, lddy);
  DMVAE_CHECK_ARG((uintptr_t)xT % 16 == 0 && (dw_table || (uintptr_t)dw % 16 == 0), "linear_rows_wgrad_batched: xT and dw must be 16-byte aligned");
  DMVAE_CHECK_ARG(layers == 1 || dw_table, "linear_rows_wgrad_batched: more than one layer needs the destination tables");
  const dim3 grid((N + 15) / 16, (K + 511) / 512, layers);
  if (mp == 32)
    hipLaunchKernelGGL(rows_wgrad_mfma_kernel<32>, grid, dim3(256), 0, stream, (const bf16*)dy, dy_layer_stride, (const bf16*)xT, (float* const*)dw_table,
                       (float* const*)db_table, (float*)dw, (float*)db, M, N, K, lddy, accumulate);
  else
    hipLaunchKernelGGL(rows_wgrad_mfma_kernel<64>, grid, dim3(256), 0, stream, (const bf16*)dy, dy_layer_stride, (const bf16*)xT, (float* const*)dw_table,
                       (float* const*)db_table, (float*)dw, (float*)db, M, N, K, lddy, accumulate);
  DMVAE_CHECK_LAUNCH();
  return 0;
}
