// Backward of the fused multi-head self-attention of vit.hip (timm Attention / dino_layers/attention.py:56-69 for the trainable ViT encoder of
// train_dmd.py:349,519; LightningDiT's Attention, diffusion/lightningdit/lightningdit.py:76-88, for the student's training turn) -- S <= 288 tokens,
// head dim 64 (ViT) or 72 zero-padded to 96 (LightningDiT-XL).  Replaces autograd's SDPA backward; the GEMM-composed version this build used before
// moved the S x S scores / probabilities through HBM four times per block.
//
//   P = softmax(scale Q K^T)        dV = P^T dO        dP = dO V^T        dS = P o (dP - delta),  delta_q = dO_q . O_q        dQ = scale dS K        dK = scale dS^T Q
//
// One workgroup (4 waves) per (batch, head), two phases, nothing of size S x S leaves the registers:
//   A  K and V resident in LDS; a wave owns 32 queries at a time (as the forward kernel): S^T = K Q^T and dP^T = V dO^T on the matrix cores with the
//      queries along the lanes (row statistics need one swap with lane ^ 32), dS^T -> bf16 A-fragments by v_permlane32_swap, dQ = dS K through the LDS
//      transpose read of K.  Leaves L_q = max + log(sum) and delta_q in LDS.
//   B  Q and dO resident in LDS (restaged over K / V); a wave owns 32 keys: S = Q K^T and dP = dO V^T with the KEYS along the lanes, P = exp(scale S - L),
//      P^T / dS^T as A-fragments by the same swap, dV += P^T dO and dK += dS^T Q through transpose reads of dO / Q.
// Seven S x S x d contractions instead of the minimal five (S and dP are formed in both orientations): the alternative is an LDS transpose of P and dS
// per 32 x 32 block or float atomics on dQ; at S <= 288 the extra matrix work is the cheaper price and every sum keeps a fixed order.
//
// LDS image of a [token][d] operand (one copy serves both access kinds): 192-B rows, 16-B chunks XOR-ed inside their 64-B segment by (row >> 2) & 3.
//   row reads (A / B fragments of the "NT" products: lane -> row l & 31, chunk 2 kk + (l >> 5)): 16 consecutive rows land in 16 distinct 16-B slots mod 256 B
//     (row * 12 mod 16 cycles through {0, 12, 8, 4}, the XOR through the other two bits);
//   transpose reads (ds_read_b64_tr_b16, B fragments with the reduction along tokens): the four rows of a pass start 48 banks apart, i.e. in four
//     different 16-bank quarters, and a pass reads one whole 64-B segment of each -- the XOR only permutes inside it.
#include "common.h"
#include "dmvae_hip.h"

namespace dmvae_attn_bwd {

constexpr int KEYS = 288, NB = KEYS / 32, PITCH = 192;
constexpr int BUF = KEYS * PITCH;

struct Args {
  const bf16 *q, *k, *v;         // per (batch, head): base + b * bs + h * hs, token rows rs elements apart; q / k rows hold DP (padded) channels
  const bf16 *o, *dout;          // [B][S][H * D]
  bf16 *dq, *dk, *dv;            // same geometry as q / k / v
  long long q_bs, q_hs, k_bs, k_hs, v_bs, v_hs;
  int q_rs, k_rs, v_rs;
  int S, H, D;
  int QD;                        // channels a q / k / dq / dk row holds: DP (zero-padded by the producer) or D itself (rows 2 D bytes apart: chunks past QD are neither loaded nor stored)
  float scale;
  const float* lse;              // attention_bwd_lse_kernel: [B * H][S] f32, scale * max + log(sum) of every query's scaled scores (the forward kernel writes it)
  int xcd;                       // 1: block -> (batch, head) through xcd_remap (vit.hip::AttnArgs::xcd)
};

__device__ __forceinline__ s16x4 tr_read(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}
__device__ __forceinline__ unsigned pack2(float a, float b) {
  bf16x2 t = {(bf16)a, (bf16)b};
  return *reinterpret_cast<unsigned*>(&t);
}
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * PITCH + (((chunk & ~3) | ((chunk & 3) ^ ((row >> 2) & 3))) << 4); }
// LDS operand layouts of the lse kernels' two block functions.  Wide: the 192-B rows above (both head widths).  Tight (72-channel heads, S <= 256 only): 144-B rows
// holding the 72 real channels, no swizzle -- row reads: sixteen rows 36 banks apart land on sixteen distinct 4-bank slots (9 is odd), conflict-free as they are;
// transpose reads: the four rows of a 16-lane group are conflict-free, the two groups of a 32-lane pass overlap in two rows (2-way).  Two images are 73.7 KB, so TWO
// workgroups fit a CU (attention_bwd_lse_tight_kernel).  The channels 72 .. 95 the 96-wide products would read are zeros in the wide image: the tight form skips the
// sixth K step, zeroes the upper half of the fifth, and leaves the (finite) garbage of output columns >= 72 of the transposed products to be dropped at the store.
struct LayWide {
  static constexpr int PITCH_ = PITCH;
  template <int DP> static constexpr int ksteps() { return DP / 16; }
  static __device__ __forceinline__ int off(int row, int chunk) { return lds_off(row, chunk); }
  static __device__ __forceinline__ bf16x8 frag(const char* buf, int row, int kk, int kg) { return *reinterpret_cast<const bf16x8*>(buf + lds_off(row, kk * 2 + kg)); }
};
struct LayTight {
  static constexpr int PITCH_ = 144;
  template <int DP> static constexpr int ksteps() { return 5; }
  static __device__ __forceinline__ int off(int row, int chunk) { return row * 144 + chunk * 16; }
  static __device__ __forceinline__ bf16x8 frag(const char* buf, int row, int kk, int kg) {
    if (kk < 4) return *reinterpret_cast<const bf16x8*>(buf + row * 144 + (kk * 2 + kg) * 16);
    bf16x8 v = *reinterpret_cast<const bf16x8*>(buf + row * 144 + 8 * 16);      // kk == 4: channels 64 .. 71 for kg == 0, nothing for kg == 1
    if (kg) {
#pragma unroll
      for (int e = 0; e < 8; e++) v[e] = (bf16)0.f;
    }
    return v;
  }
};
__device__ __forceinline__ float dot8(const uint4& a, const uint4& b) {
  const bf16x8 x = *reinterpret_cast<const bf16x8*>(&a), y = *reinterpret_cast<const bf16x8*>(&b);
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 8; e++) s += (float)x[e] * (float)y[e];
  return s;
}
// C-layout registers of a 32 x 32 block (row (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column lane & 31), 16 of them scaled to bf16 -> the two A fragments
// (reduction index = the block's ROW, 8 consecutive per lane) of its two 16-row halves
__device__ __forceinline__ void to_afrag(const f32x16& c, bf16x8 out[2]) {
#pragma unroll
  for (int half = 0; half < 2; half++) {
    const unsigned p0 = pack2(c[half * 8 + 0], c[half * 8 + 1]), p1 = pack2(c[half * 8 + 2], c[half * 8 + 3]);
    const unsigned p2 = pack2(c[half * 8 + 4], c[half * 8 + 5]), p3 = pack2(c[half * 8 + 6], c[half * 8 + 7]);
    // lanes < 32 hold rows {0-3, 8-11} of the half, lanes >= 32 rows {4-7, 12-15}: the fragment wants {0-7} / {8-15}
    const auto s0 = __builtin_amdgcn_permlane32_swap(p0, p2, false, false);
    const auto s1 = __builtin_amdgcn_permlane32_swap(p1, p3, false, false);
    union { unsigned u[4]; bf16x8 v; } pa;
    pa.u[0] = s0[0]; pa.u[1] = s1[0]; pa.u[2] = s0[1]; pa.u[3] = s1[1];
    out[half] = pa.v;
  }
}

template <int DP>
__global__ __launch_bounds__(256) void attention_bwd_kernel(Args a) {
#if __HIP_DEVICE_COMPILE__
  constexpr int KSTEPS = DP / 16, DB = DP / 32, CH = DP / 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* buf0 = smem;                    // phase A: K, phase B: Q
  char* buf1 = smem + BUF;              // phase A: V, phase B: dO
  float* Ls = reinterpret_cast<float*>(smem + 2 * BUF);  // [288] max + log(sum) of the scaled scores; +inf for padded queries
  float* Ds = Ls + KEYS;                                  // [288] delta
  const int S = a.S, H = a.H, D = a.D, C = H * D;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bf16* qg = a.q + b * a.q_bs + h * a.q_hs;
  const bf16* kg_ = a.k + b * a.k_bs + h * a.k_hs;
  const bf16* vg = a.v + b * a.v_bs + h * a.v_hs;
  const bf16* og = a.o + (size_t)b * S * C + h * D;
  const bf16* dog = a.dout + (size_t)b * S * C + h * D;
  const int dchunks = D / 8;            // real channels of v / o / dout; q and k rows carry DP (the producer zero-pads)
  const int nblk = (S + 31) >> 5;

  // stage two [token][*] operands: CH lanes x 16 B per row
  auto stage = [&](const bf16* s0, int rs0, int ch0, const bf16* s1, int rs1, int ch1) {
    for (int i = tid; i < KEYS * CH; i += 256) {
      const int row = i / CH, c = i - row * CH;
      uint4 x = {0, 0, 0, 0}, y = {0, 0, 0, 0};
      if (row < S) {
        if (c < ch0) x = *reinterpret_cast<const uint4*>(s0 + (size_t)row * rs0 + c * 8);
        if (c < ch1) y = *reinterpret_cast<const uint4*>(s1 + (size_t)row * rs1 + c * 8);
      }
      const int off = lds_off(row, c);
      *reinterpret_cast<uint4*>(buf0 + off) = x;
      *reinterpret_cast<uint4*>(buf1 + off) = y;
    }
  };
  stage(kg_, a.k_rs, a.QD / 8, vg, a.v_rs, dchunks);
  __syncthreads();

  const int kg = lane >> 5, ql = lane & 31;
  // transpose-read addressing: lane supplies 4 channels (8 B) of token row 8 kg + rr (and + 4) of a 16-token step, channel block db
  const int g16 = (lane >> 4) & 1, rr = (lane & 15) >> 2, qq = lane & 3;
  int toff0[DB], toff1[DB];
#pragma unroll
  for (int db = 0; db < DB; db++) {
    const int chunk = db * 4 + g16 * 2 + (qq >> 1);
    toff0[db] = lds_off(kg * 8 + rr, chunk) + (qq & 1) * 8;
    toff1[db] = lds_off(kg * 8 + rr + 4, chunk) + (qq & 1) * 8;
  }
  const float scale = a.scale;

  // ================= phase A: per 32-query block -- statistics, dQ ==================================================================================
  for (int qb = wave; qb < nblk; qb += 4) {
    const int q = qb * 32 + ql;
    bf16x8 qf[KSTEPS], dof[KSTEPS];
    float delta = 0.f;
#pragma unroll
    for (int kk = 0; kk < KSTEPS; kk++) {
      uint4 tq = {0, 0, 0, 0}, td = {0, 0, 0, 0}, to = {0, 0, 0, 0};
      const int d0 = kk * 16 + kg * 8;
      if (q < S) {
        if (d0 < a.QD) tq = *reinterpret_cast<const uint4*>(qg + (size_t)q * a.q_rs + d0);
        if (d0 < D) {
          td = *reinterpret_cast<const uint4*>(dog + (size_t)q * C + d0);
          to = *reinterpret_cast<const uint4*>(og + (size_t)q * C + d0);
        }
      }
      qf[kk] = *reinterpret_cast<bf16x8*>(&tq);
      dof[kk] = *reinterpret_cast<bf16x8*>(&td);
      delta += dot8(td, to);
    }
    delta += __shfl_xor(delta, 32, 64);
    // S^T = K Q^T: st[kb][r] = score(key kb*32 + (r&3) + 8 (r>>2) + 4 kg, query q)
    f32x16 st[NB];
#pragma unroll
    for (int kb = 0; kb < NB; kb++) {
#pragma unroll
      for (int r = 0; r < 16; r++) st[kb][r] = 0.f;
      const int key = kb * 32 + ql;
#pragma unroll
      for (int kk = 0; kk < KSTEPS; kk++) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(buf0 + lds_off(key, kk * 2 + kg));
        st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kk], st[kb], 0, 0, 0);
      }
    }
    float m = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < NB; kb++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
        const float v = key < S ? st[kb][r] * scale : -INFINITY;
        st[kb][r] = v;
        m = fmaxf(m, v);
      }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kb = 0; kb < NB; kb++)
#pragma unroll
      for (int r = 0; r < 16; r++) { const float e = __expf(st[kb][r] - m); st[kb][r] = e; sum += e; }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;
    if (kg == 0) { Ls[q] = q < S ? m + __logf(sum) : INFINITY; Ds[q] = delta; }
    // dP^T = V dO^T per key block, dS^T = P o (dP^T - delta) * scale, dQ += dS K
    f32x16 dq[DB];
#pragma unroll
    for (int db = 0; db < DB; db++)
#pragma unroll
      for (int r = 0; r < 16; r++) dq[db][r] = 0.f;
#pragma unroll
    for (int kb = 0; kb < NB; kb++) {
      f32x16 dpt;
#pragma unroll
      for (int r = 0; r < 16; r++) dpt[r] = 0.f;
      const int key = kb * 32 + ql;
#pragma unroll
      for (int kk = 0; kk < KSTEPS; kk++) {
        const bf16x8 vf = *reinterpret_cast<const bf16x8*>(buf1 + lds_off(key, kk * 2 + kg));
        dpt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[kk], dpt, 0, 0, 0);
      }
      const float ps = inv * scale;
#pragma unroll
      for (int r = 0; r < 16; r++) dpt[r] = st[kb][r] * ps * (dpt[r] - delta);
      bf16x8 af[2];
      to_afrag(dpt, af);
#pragma unroll
      for (int half = 0; half < 2; half++) {
        const char* base = buf0 + (kb * 2 + half) * 16 * PITCH;
#pragma unroll
        for (int db = 0; db < DB; db++) {
          union { bf16x8 v; s16x4 hlf[2]; } kf;
          kf.hlf[0] = tr_read(base + toff0[db]);
          kf.hlf[1] = tr_read(base + toff1[db]);
          dq[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[half], kf.v, dq[db], 0, 0, 0);
        }
      }
    }
    bf16* dqg = a.dq + b * a.q_bs + h * a.q_hs;
#pragma unroll
    for (int db = 0; db < DB; db++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int qo = qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
        if (qo < S && db * 32 + ql < a.QD) dqg[(size_t)qo * a.q_rs + db * 32 + ql] = (bf16)dq[db][r];
      }
  }
  __syncthreads();
  // ================= phase B: Q and dO resident; per 32-key block -- dK, dV ===========================================================================
  stage(qg, a.q_rs, a.QD / 8, dog, C, dchunks);
  __syncthreads();
  for (int kb = wave; kb < nblk; kb += 4) {
    const int key = kb * 32 + ql;
    bf16x8 kfb[KSTEPS], vfb[KSTEPS];
#pragma unroll
    for (int kk = 0; kk < KSTEPS; kk++) {
      uint4 tk = {0, 0, 0, 0}, tv = {0, 0, 0, 0};
      const int d0 = kk * 16 + kg * 8;
      if (key < S) {
        if (d0 < a.QD) tk = *reinterpret_cast<const uint4*>(kg_ + (size_t)key * a.k_rs + d0);
        if (d0 < D) tv = *reinterpret_cast<const uint4*>(vg + (size_t)key * a.v_rs + d0);
      }
      kfb[kk] = *reinterpret_cast<bf16x8*>(&tk);
      vfb[kk] = *reinterpret_cast<bf16x8*>(&tv);
    }
    f32x16 dk[DB], dv[DB];
#pragma unroll
    for (int db = 0; db < DB; db++)
#pragma unroll
      for (int r = 0; r < 16; r++) { dk[db][r] = 0.f; dv[db][r] = 0.f; }
    for (int qb = 0; qb < nblk; qb++) {
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; r++) { s[r] = 0.f; dp[r] = 0.f; }
      const int qrow = qb * 32 + ql;
#pragma unroll
      for (int kk = 0; kk < KSTEPS; kk++) {
        const int off = lds_off(qrow, kk * 2 + kg);
        const bf16x8 qfr = *reinterpret_cast<const bf16x8*>(buf0 + off);
        const bf16x8 dor = *reinterpret_cast<const bf16x8*>(buf1 + off);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr, kfb[kk], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dor, vfb[kk], dp, 0, 0, 0);
      }
      // rows: queries qb*32 + (r&3) + 8 (r>>2) + 4 kg; column: this lane's key
#pragma unroll
      for (int r4 = 0; r4 < 4; r4++) {
        const f32x4 Lv = *reinterpret_cast<const f32x4*>(Ls + qb * 32 + 8 * r4 + 4 * kg);
        const f32x4 Dv = *reinterpret_cast<const f32x4*>(Ds + qb * 32 + 8 * r4 + 4 * kg);
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const float p = key < S ? __expf(s[r4 * 4 + r] * scale - Lv[r]) : 0.f;
          s[r4 * 4 + r] = p;
          dp[r4 * 4 + r] = p * (dp[r4 * 4 + r] - Dv[r]) * scale;
        }
      }
      bf16x8 pf[2], dsf[2];
      to_afrag(s, pf);
      to_afrag(dp, dsf);
#pragma unroll
      for (int half = 0; half < 2; half++) {
        const int rbase = (qb * 2 + half) * 16 * PITCH;
#pragma unroll
        for (int db = 0; db < DB; db++) {
          union { bf16x8 v; s16x4 hlf[2]; } df, qf2;
          df.hlf[0] = tr_read(buf1 + rbase + toff0[db]);
          df.hlf[1] = tr_read(buf1 + rbase + toff1[db]);
          qf2.hlf[0] = tr_read(buf0 + rbase + toff0[db]);
          qf2.hlf[1] = tr_read(buf0 + rbase + toff1[db]);
          dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf[half], df.v, dv[db], 0, 0, 0);
          dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dsf[half], qf2.v, dk[db], 0, 0, 0);
        }
      }
    }
    bf16* dkg = a.dk + b * a.k_bs + h * a.k_hs;
    bf16* dvg = a.dv + b * a.v_bs + h * a.v_hs;
#pragma unroll
    for (int db = 0; db < DB; db++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int ko = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
        const int dcol = db * 32 + ql;
        if (ko < S) {
          if (dcol < a.QD) dkg[(size_t)ko * a.k_rs + dcol] = (bf16)dk[db][r];
          if (dcol < D) dvg[(size_t)ko * a.v_rs + dcol] = (bf16)dv[db][r];
        }
      }
  }
#endif
}

// dQ of one 32-query block (phase A of the lse kernels): K in buf0, V in buf1 (LDS images), the block's Q / dO fragments, L and delta of the lane's query.
  // A key block's LDS reads are issued in two groups ahead of their matrix products (all K / V row fragments; then, behind the S^T / dP^T products, the six K^T
  // fragments of the dQ product, which land under the softmax arithmetic): left to the scheduler, every pair of products sat behind its own LDS round trip --
  // twelve exposed latencies per key block, more than the products and the exponentials together.  Same operations on the same values: same bits.
template <int DP, class L = LayWide>
__device__ __forceinline__ void lse_dq_block(const char* buf0, const char* buf1, const bf16x8 (&qf)[L::template ksteps<DP>()], const bf16x8 (&dof)[L::template ksteps<DP>()], float Lq, float delta, float scale,
                                             int S, int nblk, int kg, int ql, const int (&toff0)[DP / 32], const int (&toff1)[DP / 32], f32x16 (&dq)[DP / 32]) {
  constexpr int KSTEPS = L::template ksteps<DP>(), DB = DP / 32;
  for (int kb = 0; kb < nblk; kb++) {
    f32x16 st, dpt;
#pragma unroll
    for (int r = 0; r < 16; r++) { st[r] = 0.f; dpt[r] = 0.f; }
    const int key = kb * 32 + ql;
    bf16x8 kf[KSTEPS], vf[KSTEPS];
#pragma unroll
    for (int kk = 0; kk < KSTEPS; kk++) {
      kf[kk] = L::frag(buf0, key, kk, kg);
      vf[kk] = L::frag(buf1, key, kk, kg);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < KSTEPS; kk++) {
      st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kk], qf[kk], st, 0, 0, 0);
      dpt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[kk], dof[kk], dpt, 0, 0, 0);
    }
    union { bf16x8 v; s16x4 hlf[2]; } ktr[2][DB];
#pragma unroll
    for (int half = 0; half < 2; half++) {
      const char* base = buf0 + (kb * 2 + half) * 16 * L::PITCH_;
#pragma unroll
      for (int db = 0; db < DB; db++) {
        ktr[half][db].hlf[0] = tr_read(base + toff0[db]);
        ktr[half][db].hlf[1] = tr_read(base + toff1[db]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (kb * 32 + 32 > S) {   // only the last block can hold keys past S (their K rows are zero: score 0, not -inf): a wave-uniform branch
#pragma unroll
      for (int r = 0; r < 16; r++) {
        float p = __expf(st[r] * scale - Lq);
        if (kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg >= S) p = 0.f;
        dpt[r] = p * (dpt[r] - delta) * scale;
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const float p = __expf(st[r] * scale - Lq);
        dpt[r] = p * (dpt[r] - delta) * scale;
      }
    }
    bf16x8 af[2];
    to_afrag(dpt, af);
#pragma unroll
    for (int half = 0; half < 2; half++)
#pragma unroll
      for (int db = 0; db < DB; db++) dq[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[half], ktr[half][db].v, dq[db], 0, 0, 0);
  }
}

// dK / dV of one 32-key block (phase B): Q in bufq, dO in bufd (LDS images), the block's K / V fragments, L and delta of every query in Ls / Ds.
template <int DP, class L = LayWide>
__device__ __forceinline__ void lse_dkdv_block(const char* bufq, const char* bufd, const float* Ls, const float* Ds, const bf16x8 (&kfb)[L::template ksteps<DP>()], const bf16x8 (&vfb)[L::template ksteps<DP>()],
                                               float scale, int S, int nblk, int kb, int kg, int ql, const int (&toff0)[DP / 32], const int (&toff1)[DP / 32],
                                               f32x16 (&dk)[DP / 32], f32x16 (&dv)[DP / 32]) {
  constexpr int KSTEPS = L::template ksteps<DP>(), DB = DP / 32;
  const int key = kb * 32 + ql;
  const bool live = kb * 32 + 32 <= S;   // every key of the block is a real one (wave-uniform): the common case takes no per-element select
  for (int qb = 0; qb < nblk; qb++) {
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; r++) { s[r] = 0.f; dp[r] = 0.f; }
    const int qrow = qb * 32 + ql;
    bf16x8 qfr[KSTEPS], dor[KSTEPS];
#pragma unroll
    for (int kk = 0; kk < KSTEPS; kk++) {
      qfr[kk] = L::frag(bufq, qrow, kk, kg);
      dor[kk] = L::frag(bufd, qrow, kk, kg);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < KSTEPS; kk++) {
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr[kk], kfb[kk], s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dor[kk], vfb[kk], dp, 0, 0, 0);
    }
    // behind the products: the row statistics of the 32 queries (under way while the products run); behind the exponentials: the transposed dO / Q fragments
    // of both 16-query halves, the first group under the bf16 packing of P / dS -- one exposed LDS latency per query block where the interleaved form had twelve
    f32x4 Lv[4], Dv[4];
#pragma unroll
    for (int r4 = 0; r4 < 4; r4++) {
      Lv[r4] = *reinterpret_cast<const f32x4*>(Ls + qb * 32 + 8 * r4 + 4 * kg);
      Dv[r4] = *reinterpret_cast<const f32x4*>(Ds + qb * 32 + 8 * r4 + 4 * kg);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r4 = 0; r4 < 4; r4++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        float p = __expf(s[r4 * 4 + r] * scale - Lv[r4][r]);
        if (!live && key >= S) p = 0.f;
        s[r4 * 4 + r] = p;
        dp[r4 * 4 + r] = p * (dp[r4 * 4 + r] - Dv[r4][r]) * scale;
      }
    union { bf16x8 v; s16x4 hlf[2]; } dft[2][DB], qft[2][DB];
#pragma unroll
    for (int db = 0; db < DB; db++) {
      const int rbase = (qb * 2) * 16 * L::PITCH_;
      dft[0][db].hlf[0] = tr_read(bufd + rbase + toff0[db]);
      dft[0][db].hlf[1] = tr_read(bufd + rbase + toff1[db]);
      qft[0][db].hlf[0] = tr_read(bufq + rbase + toff0[db]);
      qft[0][db].hlf[1] = tr_read(bufq + rbase + toff1[db]);
    }
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 pf[2], dsf[2];
    to_afrag(s, pf);
    to_afrag(dp, dsf);
#pragma unroll
    for (int db = 0; db < DB; db++) {
      const int rbase = (qb * 2 + 1) * 16 * L::PITCH_;
      dft[1][db].hlf[0] = tr_read(bufd + rbase + toff0[db]);
      dft[1][db].hlf[1] = tr_read(bufd + rbase + toff1[db]);
      qft[1][db].hlf[0] = tr_read(bufq + rbase + toff0[db]);
      qft[1][db].hlf[1] = tr_read(bufq + rbase + toff1[db]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int half = 0; half < 2; half++)
#pragma unroll
      for (int db = 0; db < DB; db++) {
        dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf[half], dft[half][db].v, dv[db], 0, 0, 0);
        dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dsf[half], qft[half][db].v, dk[db], 0, 0, 0);
      }
  }
}

// The same backward with the forward kernel's row statistics handed in (lse) and two waves per SIMD.  The kernel above keeps all nine S^T blocks of a query block
// in registers between its max / sum pass and its dS pass (144 registers; 460 in all: one wave per SIMD, so every exponential, every permlane swap and every
// LDS round trip of a wave sits between its own MFMAs -- 72 us for the 2 832 MFMAs of a (batch, head) at S = 256, D = 72: 15 % of the matrix pipe).  With
// L_q = scale max + log(sum) known, P = exp(scale s - L) needs no pass of its own: phase A walks the key blocks once with 16 + 16 score registers, a workgroup is
// eight waves -- one 32-query block each in phase A, one 32-key block each in phase B at S = 256 -- and the softmax arithmetic of one wave runs under the other's
// matrix work.  Key blocks past S are skipped (eight instead of nine at S = 256).  Same sums in the same order per output element: deterministic.
template <int DP>
__global__ __launch_bounds__(512) void attention_bwd_lse_kernel(Args a) {
#if __HIP_DEVICE_COMPILE__
  constexpr int KSTEPS = DP / 16, DB = DP / 32, CH = DP / 8, NT = 512, NW = NT / 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* buf0 = smem;                    // phase A: K, phase B: Q
  char* buf1 = smem + BUF;              // phase A: V, phase B: dO
  float* Ls = reinterpret_cast<float*>(smem + 2 * BUF);  // [288] L_q; +inf for padded queries
  float* Ds = Ls + KEYS;                                  // [288] delta
  const int S = a.S, H = a.H, D = a.D, C = H * D;
  const int bh = a.xcd ? (int)xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int b = bh / H, h = bh % H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bf16* qg = a.q + b * a.q_bs + h * a.q_hs;
  const bf16* kg_ = a.k + b * a.k_bs + h * a.k_hs;
  const bf16* vg = a.v + b * a.v_bs + h * a.v_hs;
  const bf16* og = a.o + (size_t)b * S * C + h * D;
  const bf16* dog = a.dout + (size_t)b * S * C + h * D;
  const float* lse = a.lse + (size_t)bh * S;
  const int dchunks = D / 8;
  const int nblk = (S + 31) >> 5;
  const int rows_staged = nblk * 32;

  // every global load of a staging is issued before its first LDS store (vit.hip's forward staging: the loop form waits for each sweep's loads before issuing the
  // next sweep's -- seven serial memory round trips per staging at DP = 96)
  constexpr int SWEEPS = (KEYS * CH + NT - 1) / NT;
  auto stage = [&](const bf16* s0, int rs0, int ch0, const bf16* s1, int rs1, int ch1) {
    uint4 x[SWEEPS], y[SWEEPS];
#pragma unroll
    for (int it = 0; it < SWEEPS; it++) {
      const int i = tid + it * NT, row = i / CH, c = i - row * CH;
      x[it] = uint4{0, 0, 0, 0}; y[it] = uint4{0, 0, 0, 0};
      if (row < S) {
        if (c < ch0) x[it] = *reinterpret_cast<const uint4*>(s0 + (size_t)row * rs0 + c * 8);
        if (c < ch1) y[it] = *reinterpret_cast<const uint4*>(s1 + (size_t)row * rs1 + c * 8);
      }
    }
#pragma unroll
    for (int it = 0; it < SWEEPS; it++) {
      const int i = tid + it * NT, row = i / CH, c = i - row * CH;
      if (row < rows_staged) {
        const int off = lds_off(row, c);
        *reinterpret_cast<uint4*>(buf0 + off) = x[it];
        *reinterpret_cast<uint4*>(buf1 + off) = y[it];
      }
    }
  };
  stage(kg_, a.k_rs, a.QD / 8, vg, a.v_rs, dchunks);
  for (int i = tid; i < KEYS; i += NT) Ls[i] = i < S ? lse[i] : INFINITY;
  __syncthreads();

  const int kg = lane >> 5, ql = lane & 31;
  const int g16 = (lane >> 4) & 1, rr = (lane & 15) >> 2, qq = lane & 3;
  int toff0[DB], toff1[DB];
#pragma unroll
  for (int db = 0; db < DB; db++) {
    const int chunk = db * 4 + g16 * 2 + (qq >> 1);
    toff0[db] = lds_off(kg * 8 + rr, chunk) + (qq & 1) * 8;
    toff1[db] = lds_off(kg * 8 + rr + 4, chunk) + (qq & 1) * 8;
  }
  const float scale = a.scale;

  // ================= phase A: per 32-query block -- delta, dQ ========================================================================================
  for (int qb = wave; qb < nblk; qb += NW) {
    const int q = qb * 32 + ql;
    bf16x8 qf[KSTEPS], dof[KSTEPS];
    float delta = 0.f;
#pragma unroll
    for (int kk = 0; kk < KSTEPS; kk++) {
      uint4 tq = {0, 0, 0, 0}, td = {0, 0, 0, 0}, to = {0, 0, 0, 0};
      const int d0 = kk * 16 + kg * 8;
      if (q < S) {
        if (d0 < a.QD) tq = *reinterpret_cast<const uint4*>(qg + (size_t)q * a.q_rs + d0);
        if (d0 < D) {
          td = *reinterpret_cast<const uint4*>(dog + (size_t)q * C + d0);
          to = *reinterpret_cast<const uint4*>(og + (size_t)q * C + d0);
        }
      }
      qf[kk] = *reinterpret_cast<bf16x8*>(&tq);
      dof[kk] = *reinterpret_cast<bf16x8*>(&td);
      delta += dot8(td, to);
    }
    delta += __shfl_xor(delta, 32, 64);
    if (kg == 0) Ds[q] = delta;
    const float Lq = Ls[q];                 // +inf for a padded query: every P of its column is 0
    f32x16 dq[DB];
#pragma unroll
    for (int db = 0; db < DB; db++)
#pragma unroll
      for (int r = 0; r < 16; r++) dq[db][r] = 0.f;
    lse_dq_block<DP>(buf0, buf1, qf, dof, Lq, delta, scale, S, nblk, kg, ql, toff0, toff1, dq);
    bf16* dqg = a.dq + b * a.q_bs + h * a.q_hs;
#pragma unroll
    for (int db = 0; db < DB; db++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int qo = qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
        if (qo < S && db * 32 + ql < a.QD) dqg[(size_t)qo * a.q_rs + db * 32 + ql] = (bf16)dq[db][r];
      }
  }
  // ================= phase B: Q and dO resident; per 32-key block -- dK, dV ===========================================================================
  // The wave's first key block takes its K / V fragments from the LDS image of phase A before Q / dO are staged over it (the same values the global rows hold:
  // zero rows past S, zero V channels past D) -- no second trip to memory for them; a second key block (S > 256: wave 0 only) reads global.
  bf16x8 kfb[KSTEPS], vfb[KSTEPS];
  {
    const int key = wave * 32 + ql;     // wave < NB: inside the buffers whether or not the block is live
#pragma unroll
    for (int kk = 0; kk < KSTEPS; kk++) {
      const int off = lds_off(key, kk * 2 + kg);
      kfb[kk] = *reinterpret_cast<const bf16x8*>(buf0 + off);
      vfb[kk] = *reinterpret_cast<const bf16x8*>(buf1 + off);
    }
  }
  __syncthreads();
  stage(qg, a.q_rs, a.QD / 8, dog, C, dchunks);
  __syncthreads();
  for (int kb = wave; kb < nblk; kb += NW) {
    const int key = kb * 32 + ql;
    if (kb != wave) {
#pragma unroll
      for (int kk = 0; kk < KSTEPS; kk++) {
        uint4 tk = {0, 0, 0, 0}, tv = {0, 0, 0, 0};
        const int d0 = kk * 16 + kg * 8;
        if (key < S) {
          if (d0 < a.QD) tk = *reinterpret_cast<const uint4*>(kg_ + (size_t)key * a.k_rs + d0);
          if (d0 < D) tv = *reinterpret_cast<const uint4*>(vg + (size_t)key * a.v_rs + d0);
        }
        kfb[kk] = *reinterpret_cast<bf16x8*>(&tk);
        vfb[kk] = *reinterpret_cast<bf16x8*>(&tv);
      }
    }
    f32x16 dk[DB], dv[DB];
#pragma unroll
    for (int db = 0; db < DB; db++)
#pragma unroll
      for (int r = 0; r < 16; r++) { dk[db][r] = 0.f; dv[db][r] = 0.f; }
    lse_dkdv_block<DP>(buf0, buf1, Ls, Ds, kfb, vfb, scale, S, nblk, kb, kg, ql, toff0, toff1, dk, dv);
    bf16* dkg = a.dk + b * a.k_bs + h * a.k_hs;
    bf16* dvg = a.dv + b * a.v_bs + h * a.v_hs;
#pragma unroll
    for (int db = 0; db < DB; db++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int ko = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
        const int dcol = db * 32 + ql;
        if (ko < S) {
          if (dcol < a.QD) dkg[(size_t)ko * a.k_rs + dcol] = (bf16)dk[db][r];
          if (dcol < D) dvg[(size_t)ko * a.v_rs + dcol] = (bf16)dv[db][r];
        }
      }
  }
#endif
}

// 72-channel heads, S <= 256, two or more (batch, head) items per CU: the same backward on the TIGHT layout (LayTight: 144-B rows, two images = 73.7 KB) with FOUR waves
// per workgroup, so that two workgroups are resident per CU (8 waves, two per SIMD, as before) and run out of phase: one's staging, fragment loads and stores fall
// under the other's loops -- the memory phases were 62 % of attention_bwd_lse_kernel's cycles with nothing resident to hide them.  A wave owns two 32-query blocks in
// phase A and two 32-key blocks in phase B.  Same products in the same order on the same values (the skipped K steps multiply zeros): same bits.
__global__ __launch_bounds__(256, 2) void attention_bwd_lse_tight_kernel(Args a) {
#if __HIP_DEVICE_COMPILE__
  constexpr int DP = 96, KE = 5, DB = DP / 32, CH = 9, NT = 256, NW = NT / 64, ROWS = 256, TB = ROWS * 144, SWEEPS = ROWS * CH / NT;   // KE: K steps of 16 channels that hold data
  using L = LayTight;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* buf0 = smem;                    // phase A: K, phase B: Q
  char* buf1 = smem + TB;               // phase A: V, phase B: dO
  float* Ls = reinterpret_cast<float*>(smem + 2 * TB);   // [256] L_q; +inf for padded queries
  float* Ds = Ls + ROWS;                                  // [256] delta
  const int S = a.S, H = a.H, D = a.D, C = H * D;
  const int bh = a.xcd ? (int)xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int b = bh / H, h = bh % H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bf16* qg = a.q + b * a.q_bs + h * a.q_hs;
  const bf16* kg_ = a.k + b * a.k_bs + h * a.k_hs;
  const bf16* vg = a.v + b * a.v_bs + h * a.v_hs;
  const bf16* og = a.o + (size_t)b * S * C + h * D;
  const bf16* dog = a.dout + (size_t)b * S * C + h * D;
  const float* lse = a.lse + (size_t)bh * S;
  const int nblk = (S + 31) >> 5;
  const int rows_staged = nblk * 32;

  auto stage = [&](const bf16* s0, int rs0, const bf16* s1, int rs1) {      // nine 16-B chunks (72 channels) of every row of both operands
    uint4 x[SWEEPS], y[SWEEPS];
#pragma unroll
    for (int it = 0; it < SWEEPS; it++) {
      const int i = tid + it * NT, row = i / CH, c = i - row * CH;
      x[it] = uint4{0, 0, 0, 0}; y[it] = uint4{0, 0, 0, 0};
      if (row < S) {
        x[it] = *reinterpret_cast<const uint4*>(s0 + (size_t)row * rs0 + c * 8);
        y[it] = *reinterpret_cast<const uint4*>(s1 + (size_t)row * rs1 + c * 8);
      }
    }
#pragma unroll
    for (int it = 0; it < SWEEPS; it++) {
      const int i = tid + it * NT, row = i / CH, c = i - row * CH;
      if (row < rows_staged) {
        const int off = L::off(row, c);
        *reinterpret_cast<uint4*>(buf0 + off) = x[it];
        *reinterpret_cast<uint4*>(buf1 + off) = y[it];
      }
    }
  };
  stage(kg_, a.k_rs, vg, a.v_rs);
  for (int i = tid; i < ROWS; i += NT) Ls[i] = i < S ? lse[i] : INFINITY;
  __syncthreads();

  const int kg = lane >> 5, ql = lane & 31;
  const int g16 = (lane >> 4) & 1, rr = (lane & 15) >> 2, qq = lane & 3;
  int toff0[DB], toff1[DB];
#pragma unroll
  for (int db = 0; db < DB; db++) {
    const int chunk = db * 4 + g16 * 2 + (qq >> 1);
    toff0[db] = L::off(kg * 8 + rr, chunk) + (qq & 1) * 8;
    toff1[db] = L::off(kg * 8 + rr + 4, chunk) + (qq & 1) * 8;
  }
  const float scale = a.scale;
  const bf16 zero16 = (bf16)0.f;

  // ================= phase A: per 32-query block -- delta, dQ ========================================================================================
  for (int qb = wave; qb < nblk; qb += NW) {
    const int q = qb * 32 + ql;
    bf16x8 qf[KE], dof[KE];
    float delta = 0.f;
#pragma unroll
    for (int kk = 0; kk < KE; kk++) {
      uint4 tq = {0, 0, 0, 0}, td = {0, 0, 0, 0}, to = {0, 0, 0, 0};
      const int d0 = kk * 16 + kg * 8;
      if (q < S) {
        if (d0 < a.QD) tq = *reinterpret_cast<const uint4*>(qg + (size_t)q * a.q_rs + d0);
        if (d0 < D) {
          td = *reinterpret_cast<const uint4*>(dog + (size_t)q * C + d0);
          to = *reinterpret_cast<const uint4*>(og + (size_t)q * C + d0);
        }
      }
      qf[kk] = *reinterpret_cast<bf16x8*>(&tq);
      dof[kk] = *reinterpret_cast<bf16x8*>(&td);
      delta += dot8(td, to);
    }
    delta += __shfl_xor(delta, 32, 64);
    if (kg == 0) Ds[q] = delta;
    const float Lq = Ls[q];                 // +inf for a padded query: every P of its column is 0
    f32x16 dq[DB];
#pragma unroll
    for (int db = 0; db < DB; db++)
#pragma unroll
      for (int r = 0; r < 16; r++) dq[db][r] = 0.f;
    lse_dq_block<DP, L>(buf0, buf1, qf, dof, Lq, delta, scale, S, nblk, kg, ql, toff0, toff1, dq);
    bf16* dqg = a.dq + b * a.q_bs + h * a.q_hs;
#pragma unroll
    for (int db = 0; db < DB; db++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int qo = qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
        const int dcol = db * 32 + ql;
        if (qo < S && dcol < a.QD) dqg[(size_t)qo * a.q_rs + dcol] = dcol < D ? (bf16)dq[db][r] : zero16;      // padded channels a row holds: zeros, as the wide products give them
      }
  }
  // ================= phase B: Q and dO resident; per 32-key block -- dK, dV ===========================================================================
  // The wave's first key block takes its K / V fragments from the LDS image of phase A before Q / dO are staged over it; its second block reads global.
  // (the second block's K fragments too -- 20 of the registers the tight layout leaves free; its V fragments come from memory)
  bf16x8 kfb[KE], vfb[KE], kfb2[KE];
  {
    const int key = wave * 32 + ql, key2 = key + NW * 32;      // key2 <= 255: inside the image whether or not the block is live
#pragma unroll
    for (int kk = 0; kk < KE; kk++) { kfb[kk] = L::frag(buf0, key, kk, kg); vfb[kk] = L::frag(buf1, key, kk, kg); kfb2[kk] = L::frag(buf0, key2, kk, kg); }
  }
  __syncthreads();
  stage(qg, a.q_rs, dog, C);
  __syncthreads();
  for (int kb = wave; kb < nblk; kb += NW) {
    const int key = kb * 32 + ql;
    if (kb != wave) {
#pragma unroll
      for (int kk = 0; kk < KE; kk++) {
        uint4 tv = {0, 0, 0, 0};
        const int d0 = kk * 16 + kg * 8;
        if (key < S && d0 < D) tv = *reinterpret_cast<const uint4*>(vg + (size_t)key * a.v_rs + d0);
        kfb[kk] = kfb2[kk];
        vfb[kk] = *reinterpret_cast<bf16x8*>(&tv);
      }
    }
    f32x16 dk[DB], dv[DB];
#pragma unroll
    for (int db = 0; db < DB; db++)
#pragma unroll
      for (int r = 0; r < 16; r++) { dk[db][r] = 0.f; dv[db][r] = 0.f; }
    lse_dkdv_block<DP, L>(buf0, buf1, Ls, Ds, kfb, vfb, scale, S, nblk, kb, kg, ql, toff0, toff1, dk, dv);
    bf16* dkg = a.dk + b * a.k_bs + h * a.k_hs;
    bf16* dvg = a.dv + b * a.v_bs + h * a.v_hs;
#pragma unroll
    for (int db = 0; db < DB; db++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int ko = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
        const int dcol = db * 32 + ql;
        if (ko < S) {
          if (dcol < a.QD) dkg[(size_t)ko * a.k_rs + dcol] = dcol < D ? (bf16)dk[db][r] : zero16;
          if (dcol < D) dvg[(size_t)ko * a.v_rs + dcol] = (bf16)dv[db][r];
        }
      }
  }
#endif
}

// S <= 256: the same backward with THREE resident operand images (256 rows each: 3 x 48 KiB + the row statistics = 146 KiB) and every global load issued ahead of
// the work that hides it.  s_memtime stamps of the two-buffer kernel at LightningDiT-XL/1's shape (B = 64; 86 k cycles per (batch, head)): 38 % in the two loops,
// the rest in memory phases nothing overlapped -- K / V staging 13 k, the waves' own Q / dO / O fragment loads (16-B pieces of 2 304-B rows) 15 k, the Q / dO
// staging 11 k, the stores 13 k -- with every CU in the same phase at the same time, so HBM alternates between saturated and idle.  Here: K, V and dO are staged
// together (dO once: phase A reads its fragments from the image, phase B its transposed ones), the wave's Q / O fragments are in flight under the staging, and the
// Q image of phase B is loaded into registers BEFORE phase A's loop and written over K after it.  Per (batch, head): Q twice, K / V / dO / O once (258 KB, was
// 381 KB).  Same operations on the same values as attention_bwd_lse_kernel: same bits.
constexpr int ROWS3 = 256, BUF3 = ROWS3 * PITCH;
template <int DP>
__global__ __launch_bounds__(512) void attention_bwd_lse3_kernel(Args a) {
#if __HIP_DEVICE_COMPILE__
  constexpr int KSTEPS = DP / 16, DB = DP / 32, CH = DP / 8, NT = 512, SW = (ROWS3 * CH + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* buf0 = smem;                    // phase A: K, phase B: Q
  char* buf1 = smem + BUF3;             // V
  char* buf2 = smem + 2 * BUF3;         // dO
  float* Ls = reinterpret_cast<float*>(smem + 3 * BUF3);  // [256] L_q; +inf for padded queries
  float* Ds = Ls + ROWS3;                                  // [256] delta
  const int S = a.S, H = a.H, D = a.D, C = H * D;
  const int bh = a.xcd ? (int)xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int b = bh / H, h = bh % H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kg = lane >> 5, ql = lane & 31;
  const bf16* qg = a.q + b * a.q_bs + h * a.q_hs;
  const bf16* kg_ = a.k + b * a.k_bs + h * a.k_hs;
  const bf16* vg = a.v + b * a.v_bs + h * a.v_hs;
  const bf16* og = a.o + (size_t)b * S * C + h * D;
  const bf16* dog = a.dout + (size_t)b * S * C + h * D;
  const float* lse = a.lse + (size_t)bh * S;
  const int dchunks = D / 8;
  const int nblk = (S + 31) >> 5;          // <= 8: one query block (phase A) and one key block (phase B) per wave
  const int rows_staged = nblk * 32;
  const int q = wave * 32 + ql;            // the lane's query in phase A

  // ---- every load of the first part: the three images, then the wave's own Q / O fragments ----------------------------------------------------------
  uint4 xk[SW], xv[SW], xd[SW];
#pragma unroll
  for (int it = 0; it < SW; it++) {
    const int i = tid + it * NT, row = i / CH, c = i - row * CH;
    xk[it] = uint4{0, 0, 0, 0}; xv[it] = uint4{0, 0, 0, 0}; xd[it] = uint4{0, 0, 0, 0};
    if (row < S) {
      if (c < a.QD / 8) xk[it] = *reinterpret_cast<const uint4*>(kg_ + (size_t)row * a.k_rs + c * 8);
      if (c < dchunks) {
        xv[it] = *reinterpret_cast<const uint4*>(vg + (size_t)row * a.v_rs + c * 8);
        xd[it] = *reinterpret_cast<const uint4*>(dog + (size_t)row * C + c * 8);
      }
    }
  }
  uint4 tq[KSTEPS], to[KSTEPS];
#pragma unroll
  for (int kk = 0; kk < KSTEPS; kk++) {
    const int d0 = kk * 16 + kg * 8;
    tq[kk] = uint4{0, 0, 0, 0}; to[kk] = uint4{0, 0, 0, 0};
    if (q < S) {
      if (d0 < a.QD) tq[kk] = *reinterpret_cast<const uint4*>(qg + (size_t)q * a.q_rs + d0);
      if (d0 < D) to[kk] = *reinterpret_cast<const uint4*>(og + (size_t)q * C + d0);
    }
  }
  for (int i = tid; i < ROWS3; i += NT) Ls[i] = i < S ? lse[i] : INFINITY;
#pragma unroll
  for (int it = 0; it < SW; it++) {
    const int i = tid + it * NT, row = i / CH, c = i - row * CH;
    if (row < rows_staged) {
      const int off = lds_off(row, c);
      *reinterpret_cast<uint4*>(buf0 + off) = xk[it];
      *reinterpret_cast<uint4*>(buf1 + off) = xv[it];
      *reinterpret_cast<uint4*>(buf2 + off) = xd[it];
    }
  }
  __syncthreads();
  // ---- phase B's Q image: on its way while phase A runs ------------------------------------------------------------------------------------------------
  uint4 xq[SW];
#pragma unroll
  for (int it = 0; it < SW; it++) {
    const int i = tid + it * NT, row = i / CH, c = i - row * CH;
    xq[it] = uint4{0, 0, 0, 0};
    if (row < S && c < a.QD / 8) xq[it] = *reinterpret_cast<const uint4*>(qg + (size_t)row * a.q_rs + c * 8);
  }

  const int g16 = (lane >> 4) & 1, rr = (lane & 15) >> 2, qq = lane & 3;
  int toff0[DB], toff1[DB];
#pragma unroll
  for (int db = 0; db < DB; db++) {
    const int chunk = db * 4 + g16 * 2 + (qq >> 1);
    toff0[db] = lds_off(kg * 8 + rr, chunk) + (qq & 1) * 8;
    toff1[db] = lds_off(kg * 8 + rr + 4, chunk) + (qq & 1) * 8;
  }
  const float scale = a.scale;

  // ================= phase A: the wave's 32-query block -- delta, dQ ==================================================================================
  if (wave < nblk) {
    bf16x8 qf[KSTEPS], dof[KSTEPS];
    float delta = 0.f;
#pragma unroll
    for (int kk = 0; kk < KSTEPS; kk++) {
      const uint4 td = *reinterpret_cast<const uint4*>(buf2 + lds_off(q, kk * 2 + kg));    // the dO row as staged: zero past S and past D, as the masked global load was
      qf[kk] = *reinterpret_cast<bf16x8*>(&tq[kk]);
      dof[kk] = *reinterpret_cast<const bf16x8*>(&td);
      delta += dot8(td, to[kk]);
    }
    delta += __shfl_xor(delta, 32, 64);
    if (kg == 0) Ds[q] = delta;
    const float Lq = Ls[q];
    f32x16 dq[DB];
#pragma unroll
    for (int db = 0; db < DB; db++)
#pragma unroll
      for (int r = 0; r < 16; r++) dq[db][r] = 0.f;
    lse_dq_block<DP>(buf0, buf1, qf, dof, Lq, delta, scale, S, nblk, kg, ql, toff0, toff1, dq);
    bf16* dqg = a.dq + b * a.q_bs + h * a.q_hs;
#pragma unroll
    for (int db = 0; db < DB; db++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int qo = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
        if (qo < S && db * 32 + ql < a.QD) dqg[(size_t)qo * a.q_rs + db * 32 + ql] = (bf16)dq[db][r];
      }
  }
  // ================= phase B: the wave's 32-key block -- dK, dV; its K / V fragments from the images before Q goes over K ================================
  bf16x8 kfb[KSTEPS], vfb[KSTEPS];
  {
    const int key = wave * 32 + ql;
#pragma unroll
    for (int kk = 0; kk < KSTEPS; kk++) {
      const int off = lds_off(key, kk * 2 + kg);
      kfb[kk] = *reinterpret_cast<const bf16x8*>(buf0 + off);
      vfb[kk] = *reinterpret_cast<const bf16x8*>(buf1 + off);
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < SW; it++) {
    const int i = tid + it * NT, row = i / CH, c = i - row * CH;
    if (row < rows_staged) *reinterpret_cast<uint4*>(buf0 + lds_off(row, c)) = xq[it];
  }
  __syncthreads();
  if (wave < nblk) {
    const int kb = wave;
    f32x16 dk[DB], dv[DB];
#pragma unroll
    for (int db = 0; db < DB; db++)
#pragma unroll
      for (int r = 0; r < 16; r++) { dk[db][r] = 0.f; dv[db][r] = 0.f; }
    lse_dkdv_block<DP>(buf0, buf2, Ls, Ds, kfb, vfb, scale, S, nblk, kb, kg, ql, toff0, toff1, dk, dv);
    bf16* dkg = a.dk + b * a.k_bs + h * a.k_hs;
    bf16* dvg = a.dv + b * a.v_bs + h * a.v_hs;
#pragma unroll
    for (int db = 0; db < DB; db++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int ko = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
        const int dcol = db * 32 + ql;
        if (ko < S) {
          if (dcol < a.QD) dkg[(size_t)ko * a.k_rs + dcol] = (bf16)dk[db][r];
          if (dcol < D) dvg[(size_t)ko * a.v_rs + dcol] = (bf16)dv[db][r];
        }
      }
  }
#endif
}

template <int DP>
static int launch(const Args& a, int batch, hipStream_t stream) {
  constexpr int lds = 2 * BUF + 2 * KEYS * (int)sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_bwd_kernel<DP>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done = true;
  }
  if (a.lse) {
    static bool attr2_done = false;
    if (!attr2_done) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_bwd_lse_kernel<DP>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      attr2_done = true;
    }
    static const int xcd = [] { const char* e = getenv("DMVAE_ATTN_XCD"); return !(e && e[0] == '0') ? 1 : 0; }();
    static const int three = [] { const char* e = getenv("DMVAE_ATTN_3BUF"); return !(e && e[0] == '0') ? 1 : 0; }();
    Args b_ = a;
    b_.xcd = xcd;
    // measured (profiles/r5_attention_3buf_ab.txt): 5 % faster at 256 and 512 (batch, head) blocks, 4 % slower at 1 024 -- with four rounds per CU the kernel is in
    // its bandwidth-bound regime and the larger load burst at the head of every block costs more than the hidden latency returns
    static const int tight = [] { const char* e = getenv("DMVAE_ATTN_BWD_TIGHT"); return !(e && e[0] == '0') ? 1 : 0; }();
    if constexpr (DP == 96) {
      if (tight && a.D == 72 && a.S <= 256 && batch * a.H >= 512) {     // two or more items per CU: two out-of-phase 4-wave workgroups per CU on the 144-B layout
        constexpr int ldst = 2 * 256 * 144 + 2 * 256 * (int)sizeof(float);
        static bool attrt_done = false;
        if (!attrt_done) {
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_bwd_lse_tight_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, ldst);
          attrt_done = true;
        }
        hipLaunchKernelGGL(attention_bwd_lse_tight_kernel, dim3(batch * a.H), dim3(256), ldst, stream, b_);
        DMVAE_CHECK_LAUNCH();
        return 0;
      }
    }
    if (three && a.S <= ROWS3 && batch * a.H <= 512) {
      constexpr int lds3 = 3 * BUF3 + 2 * ROWS3 * (int)sizeof(float);
      static bool attr3_done = false;
      if (!attr3_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_bwd_lse3_kernel<DP>), hipFuncAttributeMaxDynamicSharedMemorySize, lds3);
        attr3_done = true;
      }
      hipLaunchKernelGGL((attention_bwd_lse3_kernel<DP>), dim3(batch * a.H), dim3(512), lds3, stream, b_);
    } else {
      hipLaunchKernelGGL((attention_bwd_lse_kernel<DP>), dim3(batch * a.H), dim3(512), lds, stream, b_);
    }
  } else {
    hipLaunchKernelGGL((attention_bwd_kernel<DP>), dim3(batch * a.H), dim3(256), lds, stream, a);
  }
  DMVAE_CHECK_LAUNCH();
  return 0;
}

}  // namespace dmvae_attn_bwd

extern "C" int dmvae_attention_bwd_qkv_lse_bf16(const void* qkv, const void* out, const void* dout, const void* lse, void* dqkv, int batch, int seq, int heads,
                                                int head_dim, float scale, hipStream_t stream);
extern "C" int dmvae_attention_bwd_qkv_bf16(const void* qkv, const void* out, const void* dout, void* dqkv, int batch, int seq, int heads, int head_dim,
                                            float scale, hipStream_t stream) {
  return dmvae_attention_bwd_qkv_lse_bf16(qkv, out, dout, nullptr, dqkv, batch, seq, heads, head_dim, scale, stream);
}
extern "C" int dmvae_attention_bwd_qkv_lse_bf16(const void* qkv, const void* out, const void* dout, const void* lse, void* dqkv, int batch, int seq, int heads,
                                                int head_dim, float scale, hipStream_t stream) {
  using namespace dmvae_attn_bwd;
  DMVAE_CHECK_ARG(qkv && out && dout && dqkv && batch > 0 && heads > 0 && seq > 0, "attention_bwd_qkv_bf16: bad argument");
  DMVAE_CHECK_ARG(head_dim == 64 && seq <= KEYS, "attention_bwd_qkv_bf16: needs head_dim 64 and seq <= 288 (got %d, %d)", head_dim, seq);
  const long long C = (long long)heads * head_dim;
  Args a = {};
  a.q = (const bf16*)qkv; a.k = a.q + C; a.v = a.q + 2 * C;
  a.dq = (bf16*)dqkv; a.dk = a.dq + C; a.dv = a.dq + 2 * C;
  a.o = (const bf16*)out; a.dout = (const bf16*)dout;
  a.q_bs = a.k_bs = a.v_bs = (long long)seq * 3 * C; a.q_hs = a.k_hs = a.v_hs = head_dim;
  a.q_rs = a.k_rs = a.v_rs = (int)(3 * C);
  a.S = seq; a.H = heads; a.D = head_dim; a.QD = head_dim; a.scale = scale; a.lse = (const float*)lse;
  return launch<64>(a, batch, stream);
}

extern "C" int dmvae_attention_bwd_heads_lse_bf16(const void* q, const void* k, const void* v, const void* out, const void* dout, const void* lse, void* dq, void* dk,
                                                  void* dv, int batch, int seq, int heads, int head_dim, int head_dim_padded, float scale, hipStream_t stream);
extern "C" int dmvae_attention_bwd_heads_bf16(const void* q, const void* k, const void* v, const void* out, const void* dout, void* dq, void* dk, void* dv,
                                              int batch, int seq, int heads, int head_dim, int head_dim_padded, float scale, hipStream_t stream) {
  return dmvae_attention_bwd_heads_lse_bf16(q, k, v, out, dout, nullptr, dq, dk, dv, batch, seq, heads, head_dim, head_dim_padded, scale, stream);
}
extern "C" int dmvae_attention_bwd_heads_lse_bf16(const void* q, const void* k, const void* v, const void* out, const void* dout, const void* lse, void* dq, void* dk,
                                                  void* dv, int batch, int seq, int heads, int head_dim, int head_dim_padded, float scale, hipStream_t stream) {
  using namespace dmvae_attn_bwd;
  DMVAE_CHECK_ARG(q && k && v && out && dout && dq && dk && dv && batch > 0 && heads > 0 && seq > 0, "attention_bwd_heads_bf16: bad argument");
  const int dpc = (head_dim_padded + 31) / 32 * 32;      // q / k / dq / dk rows of 64 / 96 channels (zero-padded) or of head_dim channels (vit.hip: dmvae_attention_heads_lse_bf16)
  DMVAE_CHECK_ARG(seq <= KEYS && head_dim % 8 == 0 && head_dim <= head_dim_padded && (head_dim_padded == 64 || head_dim_padded == 96 || head_dim_padded == head_dim) &&
                  (dpc == 64 || dpc == 96),
                  "attention_bwd_heads_bf16: needs seq <= 288, head_dim %% 8 == 0, q / k rows of 64, 96 or head_dim <= 96 channels (got %d, %d, %d)", seq, head_dim, head_dim_padded);
  Args a = {};
  a.q = (const bf16*)q; a.k = (const bf16*)k; a.v = (const bf16*)v; a.o = (const bf16*)out; a.dout = (const bf16*)dout;
  a.dq = (bf16*)dq; a.dk = (bf16*)dk; a.dv = (bf16*)dv;
  a.q_hs = a.k_hs = (long long)seq * head_dim_padded; a.q_bs = a.k_bs = a.q_hs * heads;
  a.v_hs = (long long)seq * head_dim; a.v_bs = a.v_hs * heads;
  a.q_rs = a.k_rs = head_dim_padded; a.v_rs = head_dim;
  a.S = seq; a.H = heads; a.D = head_dim; a.QD = head_dim_padded; a.scale = scale; a.lse = (const float*)lse;
  return dpc == 64 ? launch<64>(a, batch, stream) : launch<96>(a, batch, stream);
}
