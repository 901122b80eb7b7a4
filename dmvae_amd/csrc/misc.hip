// Layout, packing and small elementwise kernels around the conv/GEMM hot path (all HBM-bound).
#include "common.h"
#include "dmvae_hip.h"

namespace dmvae_misc {

// ---- conv weight packing -------------------------------------------------------------------
// src: f32 [cout][cin][T] (PyTorch).  mode 0 (forward):  dst bf16 [cout_pad][T][cin_pad], dst[co][t][ci] = src[co][ci][t]
//                                     mode 1 (dgrad):    dst bf16 [cin_pad][T][cout_pad], dst[ci][T-1-t][co] = src[co][ci][t]
// dst2 (optional): the same values K-tile-major, dst2[col / 32][t][row][col % 32] (dmvae_conv_desc.w_layout = 1)
__global__ void pack_weight_kernel(const float* __restrict__ src, bf16* __restrict__ dst, bf16* __restrict__ dst2, int cout, int cin, int T, int rows_pad,
                                   int cols_pad, int mode) {
  const size_t total = (size_t)rows_pad * T * cols_pad;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int col = i % cols_pad;
    const size_t r = i / cols_pad;
    const int t = r % T;
    const int row = r / T;
    float v = 0.f;
    if (mode == 0) {
      if (row < cout && col < cin) v = src[((size_t)row * cin + col) * T + t];
    } else {
      if (row < cin && col < cout) v = src[((size_t)col * cin + row) * T + (T - 1 - t)];
    }
    dst[i] = (bf16)v;
    if (dst2) dst2[(((size_t)(col >> 5) * T + t) * rows_pad + row) * 32 + (col & 31)] = (bf16)v;
  }
}

// Tiled weight pack.  One block moves a 32 x 32 tile of (cout, cin) pairs with all their taps: the f32 source tile is read as 32 contiguous runs (coalesced),
// kept in LDS, and written out as 64-B runs of the bf16 operand (and 2-KiB runs of its K-tile-major copy) in either orientation.  The element-wise kernel above
// reads with a stride of T floats and scatters its second copy: 0.6 TB/s; one launch of it per weight and direction is 84 launches of 6-15 us per tokenizer step.
// A table of entries (device memory) makes it ONE launch for all of a model's stale operands after an optimiser step; entry e is one
// dmvae_pack_conv_weight_v2 call -- optionally on the sub-pixel weight WD of subpixel_weight_kernel, computed on the fly with the same summation order --
// and owns tiles [start, start + count).  Bit-identical to the element-wise kernel (same f32 source values, same rounding).
struct PackEntry {
  const float* src; bf16* dst; bf16* dst2;
  int cout, cin, T, rows_pad, cols_pad, mode, subpixel, pad0;      // cout / cin / T: of the tensor being packed (WD [cin_w][cout_w][16] for subpixel)
  unsigned long long start, count;                                 // tiles: count = ceil(rows_pad / 32) * ceil(cols_pad / 32)
};
constexpr int PACK_TMAX = 16;
__global__ __launch_bounds__(256) void pack_tiled_kernel(const PackEntry* __restrict__ tab, int n, PackEntry single) {
  extern __shared__ float S[];      // 32 rows x (32 Tm + 1) floats, Tm <= 16 (sized by the host for the largest entry)
  PackEntry e = single;
  unsigned long long tix = blockIdx.x;   // dispatch order (an XCD-aware order -- row-neighbour tiles on one XCD so that the two halves of a 128-B line meet in one L2 -- measured slower: 317 -> 422 us)
  if (n > 0) {   // the entry that owns this block's tile: the last one whose start is <= blockIdx.x
    int l = 0, r = n - 1;
    while (l < r) { const int m = (l + r + 1) >> 1; if (tab[m].start <= tix) l = m; else r = m - 1; }
    e = tab[l];
    tix -= e.start;
  }
  const int T = e.T;
  // (a, b) index the tensor being packed, [a][b][T]; the memory tensor is M[X][Y][Tm]: the same for a plain weight, W[b][a][9] for the sub-pixel form
  const int a_ext = e.mode == 0 ? e.rows_pad : e.cols_pad, b_ext = e.mode == 0 ? e.cols_pad : e.rows_pad;
  const int X = e.subpixel ? e.cin : e.cout, Y = e.subpixel ? e.cout : e.cin, Tm = e.subpixel ? 9 : T;
  const int y_ext = e.subpixel ? a_ext : b_ext;
  const int tyn = (y_ext + 31) >> 5;
  const int x0 = (int)(tix / tyn) * 32, y0 = (int)(tix % tyn) * 32;
  const int a0 = e.subpixel ? y0 : x0, b0 = e.subpixel ? x0 : y0;
  const int RS = 32 * Tm + 1;     // odd row stride: lanes running over the slow LDS index hit different banks
  const int run = 32 * Tm;        // floats of one source row inside the tile: contiguous in memory
  const float* __restrict__ src = e.src;
  // load, interior tiles whose rows are 16-B aligned (every tile of a conv weight with a multiple of 32 channels either way): eight lanes per source row, 16 B
  // per lane and round, all of a lane's rounds in flight before the first LDS store -- no index divisions (the element-wise form below spends two runtime
  // integer divisions per float, twice: the kernel was bound by them, not by its 0.9 GB)
  const bool fast = x0 + 32 <= X && y0 + 32 <= Y && ((Y * Tm) & 3) == 0 && ((run & 3) == 0) && ((reinterpret_cast<size_t>(src) & 15) == 0) && run <= 32 * PACK_TMAX;
  if (fast) {
    const int xl = threadIdx.x >> 3, l8 = threadIdx.x & 7;
    const f32x4* rowp = reinterpret_cast<const f32x4*>(src + ((size_t)(x0 + xl) * Y + y0) * Tm);
    constexpr int RMAX = (32 * PACK_TMAX / 4 + 7) / 8;      // 16 rounds of 8 x 16 B cover a 16-tap row
    f32x4 v[RMAX];
    const int n4 = run >> 2;
#pragma unroll
    for (int u = 0; u < RMAX; u++)
      if (l8 + 8 * u < n4) v[u] = rowp[l8 + 8 * u];
#pragma unroll
    for (int u = 0; u < RMAX; u++)
      if (l8 + 8 * u < n4) {
        float* d = S + xl * RS + 4 * (l8 + 8 * u);
        d[0] = v[u][0]; d[1] = v[u][1]; d[2] = v[u][2]; d[3] = v[u][3];
      }
  } else
  // load: twelve loads in flight per thread before the first LDS store (a 3x3 weight's tile is 36 floats per thread: three rounds)
  for (int j0 = threadIdx.x; j0 < 32 * run; j0 += 256 * 12) {
    constexpr int PU = 12;
    float v[PU];
#pragma unroll
    for (int u = 0; u < PU; u++) {
      const int j = j0 + u * 256;
      const int xl = j / run, rem = j - xl * run;
      const int yl = rem / Tm;
      v[u] = (j < 32 * run && x0 + xl < X && y0 + yl < Y) ? src[((size_t)(x0 + xl) * Y + y0) * Tm + rem] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < PU; u++) {
      const int j = j0 + u * 256;
      const int xl = j / run, rem = j - xl * run;
      if (j < 32 * run) S[xl * RS + rem] = v[u];
    }
  }
  __syncthreads();
  auto fetch = [&](int al, int bl, int t) -> float {
    if (!e.subpixel) return S[al * RS + bl * Tm + t];
    // WD[ci = a][co = b][r][s] = sum over the taps of W[co][ci] that land on source pixel (r, s): subpixel_weight_kernel's order
    const int sx = t & 3, r = t >> 2;
    const float* w = S + bl * RS + al * 9;
    float v = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ky++)
#pragma unroll
      for (int kx = 0; kx < 3; kx++)
        if (r + ky >= 2 && r + ky <= 3 && sx + kx >= 2 && sx + kx <= 3) v += w[ky * 3 + kx];
    return v;
  };
  // store: 4 lanes x 8 adjacent columns per (row, tap): one 16-B store per lane and copy, 64 B per row of the operand and of its K-tile-major copy.  (The first
  // form stored 4 B per lane -- 16 lanes per run: the kernel was bound by the number of store instructions, 288 per tile, not by their bytes: 315 us per
  // tokenizer step for 0.9 GB.)  Operands whose padded width is not a multiple of 8 columns keep the 2-column form below.
  if ((e.cols_pad & 7) == 0) {
    const int l4 = threadIdx.x & 3, grp = threadIdx.x >> 2;
    for (int pr = grp; pr < 32 * T; pr += 64) {
      const int rl = pr / T, to = pr - rl * T;            // local row of the operand, output tap
      const int t = e.mode == 0 ? to : T - 1 - to;        // source tap
      const int row = (e.mode == 0 ? a0 : b0) + rl, col0 = (e.mode == 0 ? b0 : a0) + 8 * l4;
      if (row >= e.rows_pad || col0 >= e.cols_pad) continue;
      bf16x8 o8;
#pragma unroll
      for (int h = 0; h < 8; h++) {
        const int cl = 8 * l4 + h;
        const int al = e.mode == 0 ? rl : cl, bl = e.mode == 0 ? cl : rl;
        const int a = a0 + al, b = b0 + bl;
        o8[h] = (bf16)((a < e.cout && b < e.cin) ? fetch(al, bl, t) : 0.f);
      }
      *reinterpret_cast<bf16x8*>(e.dst + ((size_t)row * T + to) * e.cols_pad + col0) = o8;
      if (e.dst2) *reinterpret_cast<bf16x8*>(e.dst2 + (((size_t)(col0 >> 5) * T + to) * e.rows_pad + row) * 32 + (col0 & 31)) = o8;
    }
    return;
  }
  const int l16 = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const bool pairs = (e.cols_pad & 1) == 0;
  for (int pr = grp; pr < 32 * T; pr += 16) {
    const int rl = pr / T, to = pr - rl * T;            // local row of the operand, output tap
    const int t = e.mode == 0 ? to : T - 1 - to;        // source tap
    float v2[2];
    int row = 0, col0 = 0;
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int cl = 2 * l16 + h;
      const int al = e.mode == 0 ? rl : cl, bl = e.mode == 0 ? cl : rl;
      const int a = a0 + al, b = b0 + bl;
      row = e.mode == 0 ? a : b;
      if (h == 0) col0 = e.mode == 0 ? b : a;
      v2[h] = (a < e.cout && b < e.cin) ? fetch(al, bl, t) : 0.f;
    }
    if (row >= e.rows_pad) continue;
    const size_t o = ((size_t)row * T + to) * e.cols_pad + col0;
    const size_t o2 = (((size_t)(col0 >> 5) * T + to) * e.rows_pad + row) * 32 + (col0 & 31);
    if (pairs && col0 + 1 < e.cols_pad) {
      const unsigned pk = dmvae_pack_bf16x2(v2[0], v2[1]);
      *reinterpret_cast<unsigned*>(e.dst + o) = pk;
      if (e.dst2) *reinterpret_cast<unsigned*>(e.dst2 + o2) = pk;
    } else {
#pragma unroll
      for (int h = 0; h < 2; h++)
        if (col0 + h < e.cols_pad) {
          e.dst[o + h] = (bf16)v2[h];
          if (e.dst2) e.dst2[(((size_t)((col0 + h) >> 5) * T + to) * e.rows_pad + row) * 32 + ((col0 + h) & 31)] = (bf16)v2[h];
        }
    }
  }
}
static inline size_t pack_lds_bytes(int Tm) { return (size_t)32 * (32 * Tm + 1) * sizeof(float); }
static inline void pack_attr() {   // 16 taps need 65.7 KB of dynamic LDS: above the 64-KB default
  static bool done = false;
  if (!done) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pack_tiled_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pack_lds_bytes(PACK_TMAX)); done = true; }
}

// ---- sub-pixel form of Upsample's conv (flux_ae.py:103-107: conv3x3(nearest-x2(x))) -----------------------------------
// Output pixel (2y + py, 2x + px) of the 3x3 conv over the nearest-x2 image only ever sees the 2x2 source pixels
// (y - 1 + py + a, x - 1 + px + b): taps that land on the same source pixel can be added up front.  In PyTorch terms
//   conv2d(interpolate(x, 2, 'nearest'), W, padding=1)  ==  conv_transpose2d(x, WD, stride=2, padding=1)
//   WD[ci][co][r][s] = sum_{ky in K(r)} sum_{kx in K(s)} W[co][ci][ky][kx],   K(r) = {k : 2 <= r + k <= 3}  (K(0)={2}, K(1)={1,2}, K(2)={0,1}, K(3)={0})
// -- 16 taps per SOURCE pixel instead of 9 per OUTPUT pixel: 4/9 of the multiply-adds, forward, input gradient (a 4x4 stride-2 conv over dy
// with WD) and weight gradient (that conv's weight gradient, folded back by the transpose of the map above) alike.
__global__ void subpixel_weight_kernel(const float* __restrict__ w, float* __restrict__ wd, int cout, int cin) {
  const size_t total = (size_t)cin * cout * 16;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int s = i & 3, r = (i >> 2) & 3;
    const size_t q = i >> 4;
    const int co = q % cout, ci = q / cout;
    const float* src = w + ((size_t)co * cin + ci) * 9;
    float v = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ky++)
#pragma unroll
      for (int kx = 0; kx < 3; kx++)
        if (r + ky >= 2 && r + ky <= 3 && s + kx >= 2 && s + kx <= 3) v += src[ky * 3 + kx];
    wd[i] = v;
  }
}
// dW[co][ci][ky][kx] (+)= sum_{r in {2-ky, 3-ky}} sum_{s in {2-kx, 3-kx}} dWD[ci][co][r][s]   (fixed order)
__global__ void subpixel_fold_kernel(const float* __restrict__ dwd, float* __restrict__ dw, int cout, int cin, int accumulate) {
  const size_t total = (size_t)cout * cin * 9;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int t = i % 9, ky = t / 3, kx = t - ky * 3;
    const size_t q = i / 9;
    const int ci = q % cin, co = q / cin;
    const float* src = dwd + ((size_t)ci * cout + co) * 16;
    const float v = (src[(2 - ky) * 4 + 2 - kx] + src[(2 - ky) * 4 + 3 - kx]) + (src[(3 - ky) * 4 + 2 - kx] + src[(3 - ky) * 4 + 3 - kx]);
    dw[i] = accumulate ? dw[i] + v : v;
  }
}

// ---- 2x2 sum pool (backward of nearest x2 upsample, flux_ae.py:104) ---------------------------
__global__ void sumpool2x2_kernel(const bf16* __restrict__ dy, bf16* __restrict__ dx, int N, int H, int W, int C) {
  const int c8 = C / 8;
  const size_t total = (size_t)N * H * W * c8;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int cc = i % c8;
    size_t r = i / c8;
    const int x = r % W; r /= W;
    const int y = r % H;
    const int n = r / H;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int dyy = 0; dyy < 2; dyy++)
#pragma unroll
      for (int dxx = 0; dxx < 2; dxx++) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(dy + (((size_t)n * 2 * H + 2 * y + dyy) * 2 * W + 2 * x + dxx) * C + cc * 8);
#pragma unroll
        for (int e = 0; e < 8; e++) acc[e] += (float)v[e];
      }
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; e++) o[e] = (bf16)acc[e];
    *reinterpret_cast<bf16x8*>(dx + i * 8) = o;
  }
}

// ---- 2x2 max pool on NHWC bf16 (VGG16 trunk of LPIPS, utils/lpips.py:116-153) and its backward fused with the ReLU mask ----------
__global__ void maxpool2x2_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, int N, int H, int W, int C) {
  const int c8 = C / 8;
  const size_t total = (size_t)N * H * W * c8;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int cc = i % c8;
    size_t r = i / c8;
    const int xo = r % W; r /= W;
    const int yo = r % H;
    const int n = r / H;
    const bf16* base = x + (((size_t)n * 2 * H + 2 * yo) * 2 * W + 2 * xo) * C + cc * 8;
    bf16x8 m = dmvae_ldnt8(base);
#pragma unroll
    for (int k = 1; k < 4; k++) {
      const bf16x8 v = dmvae_ldnt8(base + ((size_t)(k >> 1) * 2 * W + (k & 1)) * C);
#pragma unroll
      for (int e = 0; e < 8; e++) m[e] = (float)v[e] > (float)m[e] ? v[e] : m[e];
    }
    *reinterpret_cast<bf16x8*>(y + i * 8) = m;
  }
}
// dx[pos] = x[pos] > 0 ? (pos == argmax of its window ? dpool : 0) + extra[pos] : 0, argmax = first maximum in scan order
// (F.max_pool2d backward), x = the post-ReLU activation that was pooled.  dpool / extra may be null.
__global__ void maxpool2x2_relu_bwd_kernel(const bf16* __restrict__ dpool, const bf16* __restrict__ x, const bf16* __restrict__ extra,
                                           bf16* __restrict__ dx, int N, int H, int W, int C) {
  const int c8 = C / 8;
  const size_t total = (size_t)N * H * W * c8;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int cc = i % c8;
    size_t r = i / c8;
    const int xo = r % W; r /= W;
    const int yo = r % H;
    const int n = r / H;
    const size_t b0 = (((size_t)n * 2 * H + 2 * yo) * 2 * W + 2 * xo) * C + cc * 8;
    bf16x8 v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = dmvae_ldnt8(x + b0 + ((size_t)(k >> 1) * 2 * W + (k & 1)) * C);
    bf16x8 g;
#pragma unroll
    for (int e = 0; e < 8; e++) g[e] = (bf16)0.f;
    if (dpool) g = dmvae_ldnt8(dpool + i * 8);
    int arg[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
      float m = (float)v[0][e];
      arg[e] = 0;
#pragma unroll
      for (int k = 1; k < 4; k++)
        if ((float)v[k][e] > m) { m = (float)v[k][e]; arg[e] = k; }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const size_t off = b0 + ((size_t)(k >> 1) * 2 * W + (k & 1)) * C;
      bf16x8 ex;
#pragma unroll
      for (int e = 0; e < 8; e++) ex[e] = (bf16)0.f;
      if (extra) ex = *reinterpret_cast<const bf16x8*>(extra + off);
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const float t = (arg[e] == k ? (float)g[e] : 0.f) + (float)ex[e];
        o[e] = (float)v[k][e] > 0.f ? (bf16)t : (bf16)0.f;
      }
      *reinterpret_cast<bf16x8*>(dx + off) = o;
    }
  }
}
// dx = y > 0 ? dy : 0 (ReLU backward from the saved output), bf16
__global__ void relu_bwd_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ y, bf16* __restrict__ dx, size_t n8, float slope) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    const bf16x8 d = reinterpret_cast<const bf16x8*>(dy)[i], v = reinterpret_cast<const bf16x8*>(y)[i];
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; e++) o[e] = (float)v[e] > 0.f ? d[e] : (bf16)(slope * (float)d[e]);
    reinterpret_cast<bf16x8*>(dx)[i] = o;
  }
}

// ---- im2col / col2im for the PatchGAN convs (models/patchgan.py:125-147: 4x4, stride 2 or 1, padding 1) --------------------------
// col[n, oy, ox, (ky*ks + kx)*C + c] = x[n, oy*stride - pad + ky, ox*stride - pad + kx, c] (0 outside): the conv becomes the
// [M, ks*ks*C] x [Cout, ks*ks*C]^T GEMM of the 1x1 path.  One thread per (output pixel, tap, 8 channels): 16-B loads and stores.
// Tp >= ks * ks: taps past the last one are columns of zeros (the reduction dimension padded to what the consumer's tile wants).
// Cs >= C: channel stride of the SOURCE pixels (the first C of Cs channels are taken: the PatchGAN's first layer travels zero-padded to 32 channels, its weight
// gradient wants an 8-channel im2col -- without a sliced copy of the input in between).
__global__ void im2col_kernel(const bf16* __restrict__ x, bf16* __restrict__ col, int N, int H, int W, int C, int Cs, int Ho, int Wo, int ks,
                              int stride, int pad, int Tp) {
  const int c8 = C / 8, T = Tp, Treal = ks * ks;
  const size_t total = (size_t)N * Ho * Wo * T * c8;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int cc = i % c8;
    size_t r = i / c8;
    const int t = r % T; r /= T;
    const int ox = r % Wo; r /= Wo;
    const int oy = r % Ho;
    const int n = r / Ho;
    const int iy = oy * stride - pad + t / ks, ix = ox * stride - pad + t % ks;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = (bf16)0.f;
    if (t < Treal && iy >= 0 && iy < H && ix >= 0 && ix < W) v = *reinterpret_cast<const bf16x8*>(x + (((size_t)n * H + iy) * W + ix) * Cs + cc * 8);
    reinterpret_cast<bf16x8*>(col)[i] = v;
  }
}

// Adjoint as a gather (deterministic): dx[n, iy, ix, c] = sum over the taps (ky, kx) whose window covers (iy, ix) of
// dcol[n, (iy + pad - ky)/stride, (ix + pad - kx)/stride, (ky*ks + kx)*C + c]; f32 accumulation, one bf16 rounding.
template <typename TIN>
__global__ void col2im_kernel(const TIN* __restrict__ dcol, bf16* __restrict__ dx, int N, int H, int W, int C, int Ho, int Wo, int ks,
                              int stride, int pad) {
  const int c8 = C / 8, T = ks * ks;
  const size_t total = (size_t)N * H * W * c8;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int cc = i % c8;
    size_t r = i / c8;
    const int ix = r % W; r /= W;
    const int iy = r % H;
    const int n = r / H;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int ky = 0; ky < ks; ky++) {
      const int ty = iy + pad - ky;
      if (ty < 0 || ty % stride != 0 || ty / stride >= Ho) continue;
      for (int kx = 0; kx < ks; kx++) {
        const int tx = ix + pad - kx;
        if (tx < 0 || tx % stride != 0 || tx / stride >= Wo) continue;
        const TIN* src = dcol + ((((size_t)n * Ho + ty / stride) * Wo + tx / stride) * T + ky * ks + kx) * C + cc * 8;
        if constexpr (sizeof(TIN) == 4) {
          const f32x4 lo = *reinterpret_cast<const f32x4*>(src), hi = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
          for (int e = 0; e < 4; e++) { acc[e] += lo[e]; acc[4 + e] += hi[e]; }
        } else {
          const bf16x8 v = *reinterpret_cast<const bf16x8*>(src);
#pragma unroll
          for (int e = 0; e < 8; e++) acc[e] += (float)v[e];
        }
      }
    }
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; e++) o[e] = (bf16)acc[e];
    reinterpret_cast<bf16x8*>(dx)[i] = o;
  }
}

// ---- image layout: NCHW f32 <-> NHWC (channel-padded) ----------------------------------------
// one thread per (pixel, 8-channel chunk): plane reads are coalesced across pixels, the 16-B store is contiguous in NHWC
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, bf16* __restrict__ dst, int N, int C, int HW, int Cpad) {
  const int c8 = Cpad / 8;
  const size_t total = (size_t)N * HW * c8;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % c8) * 8;
    const size_t px = i / c8;
    const size_t n = px / HW, p = px % HW;
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; e++) o[e] = (bf16)(ch + e < C ? src[(n * C + ch + e) * HW + p] : 0.f);
    *reinterpret_cast<bf16x8*>(dst + px * Cpad + ch) = o;
  }
}
__global__ void nchw_to_nhwc_scalar_kernel(const float* __restrict__ src, bf16* __restrict__ dst, int N, int C, int HW, int Cpad) {
  const size_t total = (size_t)N * HW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t n = i / HW, p = i % HW;
    for (int c = 0; c < Cpad; c++) dst[i * Cpad + c] = (bf16)(c < C ? src[(n * C + c) * HW + p] : 0.f);
  }
}
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ src, float* __restrict__ dst, int N, int C, int HW, int Cpad) {
  const size_t total = (size_t)N * HW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t n = i / HW, p = i % HW;
    for (int c = 0; c < C; c++) dst[(n * C + c) * HW + p] = (float)src[i * Cpad + c];
  }
}

// ---- SiLU on bf16 (vae.py:60) ------------------------------------------------------------------
__global__ void silu_fwd_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, size_t n8) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    const bf16x8 v = reinterpret_cast<const bf16x8*>(x)[i];
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; e++) { const float t = (float)v[e]; o[e] = (bf16)(t * sigmoidf_(t)); }
    reinterpret_cast<bf16x8*>(y)[i] = o;
  }
}
__global__ void silu_bwd_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dy, bf16* __restrict__ dx, size_t n8) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    const bf16x8 v = reinterpret_cast<const bf16x8*>(x)[i];
    const bf16x8 d = reinterpret_cast<const bf16x8*>(dy)[i];
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const float t = (float)v[e], sg = sigmoidf_(t);
      o[e] = (bf16)((float)d[e] * sg * (1.f + t * (1.f - sg)));
    }
    reinterpret_cast<bf16x8*>(dx)[i] = o;
  }
}

// ---- row softmax for the decoder self-attention (flux_ae.py:47, S=1024) ------------------------
// P[r][:] = softmax(scale * S[r][:]) ; one wave per row, f32 in, bf16 out.
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ s, bf16* __restrict__ p, int rows, int cols,
                                                          float scale) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* sr = s + (size_t)row * cols;
  float m = -INFINITY;
  for (int c = lane; c < cols; c += 64) m = fmaxf(m, sr[c] * scale);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  float sum = 0.f;
  for (int c = lane; c < cols; c += 64) sum += __expf(sr[c] * scale - m);
  sum = wave_sum(sum);
  const float inv = 1.f / sum;
  bf16* pr = p + (size_t)row * cols;
  for (int c = lane; c < cols; c += 64) pr[c] = (bf16)(__expf(sr[c] * scale - m) * inv);
}
// The same with the row in registers (cols = 256 NV, NV <= 8: the decoder's S = 1024): lane l holds columns 4 l + 256 k + 0..3 -- 16-B loads, ONE exponential per
// element, 8-B stores.  (The three-pass form above reads the row three times with 4-B loads, takes three exponentials per element and stores 2 B per lane:
// 77 us for 32 x 1024 x 1024, 2.6 TB/s.)  Per-lane partial sums in column order, then the wave tree: a different summation order from the form above.
template <int NV>
__global__ __launch_bounds__(256) void softmax_fwd_rows_kernel(const float* __restrict__ s, bf16* __restrict__ p, int rows, float scale) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const f32x4* sr = reinterpret_cast<const f32x4*>(s + (size_t)row * (256 * NV)) + lane;
  f32x4 v[NV];
#pragma unroll
  for (int k = 0; k < NV; k++) v[k] = __builtin_nontemporal_load(sr + 64 * k);
  float m = -INFINITY;
#pragma unroll
  for (int k = 0; k < NV; k++) {
    v[k] *= scale;
    m = fmaxf(m, fmaxf(fmaxf(v[k][0], v[k][1]), fmaxf(v[k][2], v[k][3])));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < NV; k++)
#pragma unroll
    for (int i = 0; i < 4; i++) { v[k][i] = __expf(v[k][i] - m); sum += v[k][i]; }
  sum = wave_sum(sum);
  const float inv = 1.f / sum;
  bf16x4* pr = reinterpret_cast<bf16x4*>(p + (size_t)row * (256 * NV)) + lane;
#pragma unroll
  for (int k = 0; k < NV; k++) pr[64 * k] = bf16x4{(bf16)(v[k][0] * inv), (bf16)(v[k][1] * inv), (bf16)(v[k][2] * inv), (bf16)(v[k][3] * inv)};
}
template <int NV>
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const float* __restrict__ dp, const bf16* __restrict__ p, bf16* __restrict__ ds, int rows, float scale) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const f32x4* dr = reinterpret_cast<const f32x4*>(dp + (size_t)row * (256 * NV)) + lane;
  const bf16x4* pr = reinterpret_cast<const bf16x4*>(p + (size_t)row * (256 * NV)) + lane;
  f32x4 d[NV];
  bf16x4 q[NV];
#pragma unroll
  for (int k = 0; k < NV; k++) { d[k] = __builtin_nontemporal_load(dr + 64 * k); q[k] = __builtin_nontemporal_load(pr + 64 * k); }
  float dot = 0.f;
#pragma unroll
  for (int k = 0; k < NV; k++)
#pragma unroll
    for (int i = 0; i < 4; i++) dot += d[k][i] * (float)q[k][i];
  dot = wave_sum(dot);
  bf16x4* o = reinterpret_cast<bf16x4*>(ds + (size_t)row * (256 * NV)) + lane;
#pragma unroll
  for (int k = 0; k < NV; k++)
    o[64 * k] = bf16x4{(bf16)(scale * (float)q[k][0] * (d[k][0] - dot)), (bf16)(scale * (float)q[k][1] * (d[k][1] - dot)), (bf16)(scale * (float)q[k][2] * (d[k][2] - dot)),
                       (bf16)(scale * (float)q[k][3] * (d[k][3] - dot))};
}
// dS[r][:] = scale * P .* (dP - sum(dP .* P)) ; dP f32, P bf16, dS bf16
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ dp, const bf16* __restrict__ p,
                                                          bf16* __restrict__ ds, int rows, int cols, float scale) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* dr = dp + (size_t)row * cols;
  const bf16* pr = p + (size_t)row * cols;
  float dot = 0.f;
  for (int c = lane; c < cols; c += 64) dot += dr[c] * (float)pr[c];
  dot = wave_sum(dot);
  bf16* o = ds + (size_t)row * cols;
  for (int c = lane; c < cols; c += 64) o[c] = (bf16)(scale * (float)pr[c] * (dr[c] - dot));
}

// ---- bf16 [rows][cols] -> [cols][rows] batched transpose (attention K/V operands) ----------------
__global__ void transpose_kernel(const bf16* __restrict__ src, bf16* __restrict__ dst, int rows, int cols) {
  __shared__ bf16 tile[32][33];
  src += (size_t)blockIdx.z * rows * cols;
  dst += (size_t)blockIdx.z * rows * cols;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int j = threadIdx.y; j < 32; j += 8) {
    const int r = r0 + j, c = c0 + threadIdx.x;
    if (r < rows && c < cols) tile[j][threadIdx.x] = src[(size_t)r * cols + c];
  }
  __syncthreads();
  for (int j = threadIdx.y; j < 32; j += 8) {
    const int c = c0 + j, r = r0 + threadIdx.x;
    if (r < rows && c < cols) dst[(size_t)c * rows + r] = tile[threadIdx.x][j];
  }
}

// ---- Linear weight, bf16 [N][K] row-major -> the K-tile-major operand of its TRANSPOSE: dst[N / 32][K][32], dst[(n >> 5)][k][n & 31] = src[n][k] ------------------
// (the input-gradient operand of csrc/gemm_pp.hip: dX = dY . W is an NT GEMM whose "weight" is W^T [K][N] with the reduction over n; w_layout = 1 of that
// operand is exactly this array).  One 32-row band of src per block row: a band's result is a K x 32 transpose.  256 threads: a 32 x 64 tile through LDS,
// 16-B loads (8 columns of one row), 16-B stores (8 rows of one column ... i.e. 8 consecutive n of one k).  Replaces the element-wise f32 pack
// (35 us per 7-M-element weight: strided reads, 2-byte scattered writes) on the trainable routes, where every weight needs the copy once per step.
__global__ __launch_bounds__(256) void linear_wt_kmajor_kernel(const bf16* __restrict__ src, bf16* __restrict__ dst, int N, int K) {
  __shared__ __attribute__((aligned(16))) bf16 tile[32][64 + 8];     // +8: 16-B-aligned rows, the column reads below then hit distinct banks
  const int n0 = blockIdx.y * 32, k0 = blockIdx.x * 64;
  {
    const int r = threadIdx.x >> 3, c8 = (threadIdx.x & 7) * 8;      // 32 rows x 8 chunks of 8 columns
    uint4 v = {0u, 0u, 0u, 0u};
    if (n0 + r < N && k0 + c8 < K) v = *reinterpret_cast<const uint4*>(src + (size_t)(n0 + r) * K + k0 + c8);
    *reinterpret_cast<uint4*>(&tile[r][c8]) = v;
  }
  __syncthreads();
  {
    const int k = threadIdx.x >> 2, r8 = (threadIdx.x & 3) * 8;      // 64 columns (k) x 4 chunks of 8 rows (n)
    if (k0 + k < K) {
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; e++) o[e] = tile[r8 + e][k];
      *reinterpret_cast<bf16x8*>(dst + ((size_t)blockIdx.y * K + k0 + k) * 32 + r8) = o;
    }
  }
}

// The same transpose for a TABLE of weights in one launch (every Linear weight of a trainable transformer after its optimiser step: 142 launches of ~6.6 us for
// LightningDiT-XL/1 before).  Entry e: tiles [start_e, start_{e+1}) of the flat grid, tile = (64-column block bx, 32-row band by) with bx fastest.
struct WtEntry { const bf16* src; bf16* dst; int N, K; unsigned start, tiles_x; };
__global__ __launch_bounds__(256) void linear_wt_kmajor_batched_kernel(const WtEntry* __restrict__ tab, int n) {
  __shared__ __attribute__((aligned(16))) bf16 tile[32][64 + 8];
  int lo = 0, hi = n - 1;
  while (lo < hi) {   // last entry whose start <= blockIdx.x (block-uniform: scalar loads)
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].start <= blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const WtEntry e = tab[lo];
  const unsigned tl = blockIdx.x - e.start;
  const int by = (int)(tl / e.tiles_x), bx = (int)(tl - (unsigned)by * e.tiles_x);
  const int n0 = by * 32, k0 = bx * 64, N = e.N, K = e.K;
  {
    const int r = threadIdx.x >> 3, c8 = (threadIdx.x & 7) * 8;
    uint4 v = {0u, 0u, 0u, 0u};
    if (n0 + r < N && k0 + c8 < K) v = *reinterpret_cast<const uint4*>(e.src + (size_t)(n0 + r) * K + k0 + c8);
    *reinterpret_cast<uint4*>(&tile[r][c8]) = v;
  }
  __syncthreads();
  {
    const int k = threadIdx.x >> 2, r8 = (threadIdx.x & 3) * 8;
    if (k0 + k < K) {
      bf16x8 o;
#pragma unroll
      for (int q = 0; q < 8; q++) o[q] = tile[r8 + q][k];
      *reinterpret_cast<bf16x8*>(e.dst + ((size_t)by * K + k0 + k) * 32 + r8) = o;
    }
  }
}

static inline int grid_for(size_t n, int block = 256, int cap = 4096) {
  size_t g = (n + block - 1) / block;
  return (int)(g > (size_t)cap ? cap : (g < 1 ? 1 : g));
}

}  // namespace dmvae_misc
using namespace dmvae_misc;

extern "C" int dmvae_pack_conv_weight_v2(const void* w, void* out, void* out_kmajor, int cout, int cin, int ks, int rows_pad, int cols_pad,
                                         int for_dgrad, hipStream_t stream);
extern "C" int dmvae_pack_conv_weight(const void* w, void* out, int cout, int cin, int ks, int rows_pad, int cols_pad,
                                      int for_dgrad, hipStream_t stream) {
  return dmvae_pack_conv_weight_v2(w, out, nullptr, cout, cin, ks, rows_pad, cols_pad, for_dgrad, stream);
}
extern "C" int dmvae_pack_conv_weight_v2(const void* w, void* out, void* out_kmajor, int cout, int cin, int ks, int rows_pad, int cols_pad,
                                         int for_dgrad, hipStream_t stream) {
  DMVAE_CHECK_ARG(w && out && cout > 0 && cin > 0 && ks >= 1 && ks <= 7, "pack_conv_weight: bad argument");
  DMVAE_CHECK_ARG(!out_kmajor || cols_pad % 32 == 0, "pack_conv_weight: the K-tile-major copy needs a multiple of 32 columns (got %d)", cols_pad);
  DMVAE_CHECK_ARG(rows_pad >= (for_dgrad ? cin : cout) && cols_pad >= (for_dgrad ? cout : cin), "pack_conv_weight: padding smaller than shape");
  const int T = ks * ks;
  const size_t total = (size_t)rows_pad * T * cols_pad;
  // single-weight calls stay on the element-wise kernel: the tiled kernel (the one-launch table, dmvae_pack_weights_batched) measured 1.5 x slower here -- a
  // 512 x 512 weight is 256 tiles, one block per CU, each a serial load -> store; the element-wise kernel spreads the same weight over 9 x as many blocks
  {
    hipLaunchKernelGGL(pack_weight_kernel, dim3(grid_for(total)), dim3(256), 0, stream, (const float*)w, (bf16*)out, (bf16*)out_kmajor, cout, cin, T, rows_pad,
                       cols_pad, for_dgrad ? 1 : 0);
  }
  DMVAE_CHECK_LAUNCH();
  return 0;
}

extern "C" size_t dmvae_pack_entry_bytes(void) { return sizeof(PackEntry); }
extern "C" int dmvae_pack_weights_batched(const void* table, int n_entries, unsigned long long total_tiles, int max_taps, hipStream_t stream) {
  DMVAE_CHECK_ARG(table && n_entries > 0 && total_tiles > 0 && total_tiles < (1ull << 31), "pack_weights_batched: empty or oversized table");
  DMVAE_CHECK_ARG(max_taps >= 1 && max_taps <= PACK_TMAX, "pack_weights_batched: max_taps (the largest ks * ks of the table; 9 for a sub-pixel entry) must be 1 .. 16");
  PackEntry none = {};
  pack_attr();
  hipLaunchKernelGGL(pack_tiled_kernel, dim3((unsigned)total_tiles), dim3(256), pack_lds_bytes(max_taps), stream, (const PackEntry*)table, n_entries, none);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

extern "C" int dmvae_linear_weight_t_kmajor(const void* w, void* out, int N, int K, hipStream_t stream) {
  DMVAE_CHECK_ARG(w && out && N > 0 && K > 0 && N % 32 == 0 && K % 8 == 0, "linear_weight_t_kmajor: need N %% 32 == 0 and K %% 8 == 0 (N %d, K %d)", N, K);
  hipLaunchKernelGGL(linear_wt_kmajor_kernel, dim3((K + 63) / 64, N / 32), dim3(256), 0, stream, (const bf16*)w, (bf16*)out, N, K);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

// nn.BatchNorm2d's running-estimate update from this batch's (mean, rstd) pairs: running_mean = (1 - m) running_mean + m mean, running_var = (1 - m) running_var +
// m unbias var with var = max(1 / rstd^2 - eps, 0) -- one launch for the nine ATen launches per BatchNorm layer and discriminator pass (models/patchgan.py).
__global__ void bn_running_update_kernel(const float* __restrict__ st, float* __restrict__ rm, float* __restrict__ rv, int C, float eps, float mom, float unbias) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float mean = st[2 * c], rstd = st[2 * c + 1];
  const float var = fmaxf(1.0f / (rstd * rstd) - eps, 0.f);
  rm[c] = rm[c] * (1.f - mom) + mean * mom;
  rv[c] = rv[c] * (1.f - mom) + (var * unbias) * mom;
}
extern "C" int dmvae_batchnorm_running_update(const void* stats, void* running_mean, void* running_var, int c, float eps, float momentum, float unbias,
                                              hipStream_t stream) {
  DMVAE_CHECK_ARG(stats && running_mean && running_var && c > 0, "batchnorm_running_update: bad argument");
  hipLaunchKernelGGL(bn_running_update_kernel, dim3((c + 255) / 256), dim3(256), 0, stream, (const float*)stats, (float*)running_mean, (float*)running_var, c, eps,
                     momentum, unbias);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

extern "C" size_t dmvae_wt_entry_bytes(void) { return sizeof(WtEntry); }
extern "C" int dmvae_linear_weight_t_kmajor_batched(const void* table, int n_entries, unsigned total_tiles, hipStream_t stream) {
  DMVAE_CHECK_ARG(table && n_entries > 0 && total_tiles > 0, "linear_weight_t_kmajor_batched: empty table");
  hipLaunchKernelGGL(linear_wt_kmajor_batched_kernel, dim3(total_tiles), dim3(256), 0, stream, (const WtEntry*)table, n_entries);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

extern "C" int dmvae_subpixel_weight(const void* w, void* wd, int cout, int cin, hipStream_t stream) {
  DMVAE_CHECK_ARG(w && wd && cout > 0 && cin > 0, "subpixel_weight: bad argument");
  hipLaunchKernelGGL(subpixel_weight_kernel, dim3(grid_for((size_t)cin * cout * 16)), dim3(256), 0, stream, (const float*)w, (float*)wd, cout, cin);
  DMVAE_CHECK_LAUNCH();
  return 0;
}
extern "C" int dmvae_subpixel_weight_fold(const void* dwd, void* dw, int cout, int cin, int accumulate, hipStream_t stream) {
  DMVAE_CHECK_ARG(dwd && dw && cout > 0 && cin > 0, "subpixel_weight_fold: bad argument");
  hipLaunchKernelGGL(subpixel_fold_kernel, dim3(grid_for((size_t)cin * cout * 9)), dim3(256), 0, stream, (const float*)dwd, (float*)dw, cout, cin, accumulate);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

extern "C" int dmvae_sumpool2x2_nhwc(const void* dy, void* dx, int n, int h, int w, int c, hipStream_t stream) {
  DMVAE_CHECK_ARG(dy && dx && n > 0 && h > 0 && w > 0 && c > 0 && c % 8 == 0, "sumpool2x2_nhwc: bad argument (c must be a multiple of 8)");
  hipLaunchKernelGGL(sumpool2x2_kernel, dim3(grid_for((size_t)n * h * w * (c / 8))), dim3(256), 0, stream, (const bf16*)dy, (bf16*)dx, n, h, w, c);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

extern "C" int dmvae_maxpool2x2_nhwc(const void* x, void* y, int n, int h, int w, int c, hipStream_t stream) {
  DMVAE_CHECK_ARG(x && y && n > 0 && h > 0 && w > 0 && c > 0 && c % 8 == 0, "maxpool2x2_nhwc: bad argument (c must be a multiple of 8)");
  hipLaunchKernelGGL(maxpool2x2_kernel, dim3(grid_for((size_t)n * h * w * (c / 8))), dim3(256), 0, stream, (const bf16*)x, (bf16*)y, n, h, w, c);
  DMVAE_CHECK_LAUNCH();
  return 0;
}
extern "C" int dmvae_maxpool2x2_relu_bwd_nhwc(const void* dpool, const void* x, const void* extra, void* dx, int n, int h, int w, int c,
                                              hipStream_t stream) {
  DMVAE_CHECK_ARG(x && dx && n > 0 && h > 0 && w > 0 && c > 0 && c % 8 == 0, "maxpool2x2_relu_bwd_nhwc: bad argument (c must be a multiple of 8)");
  hipLaunchKernelGGL(maxpool2x2_relu_bwd_kernel, dim3(grid_for((size_t)n * h * w * (c / 8))), dim3(256), 0, stream, (const bf16*)dpool,
                     (const bf16*)x, (const bf16*)extra, (bf16*)dx, n, h, w, c);
  DMVAE_CHECK_LAUNCH();
  return 0;
}
extern "C" int dmvae_relu_bwd(const void* dy, const void* y, void* dx, size_t n, hipStream_t stream) {
  DMVAE_CHECK_ARG(dy && y && dx && n % 8 == 0, "relu_bwd: element count must be a multiple of 8");
  if (n == 0) return 0;
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for(n / 8)), dim3(256), 0, stream, (const bf16*)dy, (const bf16*)y, (bf16*)dx, n / 8, 0.f);
  DMVAE_CHECK_LAUNCH();
  return 0;
}
extern "C" int dmvae_leaky_relu_bwd(const void* dy, const void* y, void* dx, size_t n, float slope, hipStream_t stream) {
  DMVAE_CHECK_ARG(dy && y && dx && n % 8 == 0 && slope >= 0.f, "leaky_relu_bwd: element count must be a multiple of 8, slope >= 0");
  if (n == 0) return 0;
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for(n / 8)), dim3(256), 0, stream, (const bf16*)dy, (const bf16*)y, (bf16*)dx, n / 8, slope);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

static bool im2col_geom(int h, int w, int ks, int stride, int pad, int* ho, int* wo) {
  if (ks < 1 || ks > 7 || stride < 1 || stride > 4 || pad < 0 || pad >= ks) return false;
  *ho = (h + 2 * pad - ks) / stride + 1; *wo = (w + 2 * pad - ks) / stride + 1;
  return h + 2 * pad >= ks && w + 2 * pad >= ks;
}
extern "C" int dmvae_im2col_nhwc_taps(const void* x, void* col, int n, int h, int w, int c, int ks, int stride, int pad, int taps_pad, hipStream_t stream) {
  return dmvae_im2col_nhwc_sub(x, col, n, h, w, c, c, ks, stride, pad, taps_pad, stream);
}
extern "C" int dmvae_im2col_nhwc_sub(const void* x, void* col, int n, int h, int w, int c_src, int c, int ks, int stride, int pad, int taps_pad, hipStream_t stream) {
  int ho, wo;
  DMVAE_CHECK_ARG(c_src >= c && c_src % 8 == 0, "im2col_nhwc_sub: the source's %d channels must be a multiple of 8 and at least the %d taken", c_src, c);
  DMVAE_CHECK_ARG(x && col && n > 0 && h > 0 && w > 0 && c > 0 && c % 8 == 0 && im2col_geom(h, w, ks, stride, pad, &ho, &wo),
                  "im2col_nhwc: bad argument (c must be a multiple of 8; ks 1..7, stride 1..4, pad < ks)");
  DMVAE_CHECK_ARG(taps_pad >= ks * ks && taps_pad <= 64, "im2col_nhwc: taps_pad %d below ks * ks = %d (or above 64)", taps_pad, ks * ks);
  hipLaunchKernelGGL(im2col_kernel, dim3(grid_for((size_t)n * ho * wo * taps_pad * (c / 8))), dim3(256), 0, stream, (const bf16*)x, (bf16*)col, n, h, w, c,
                     c_src, ho, wo, ks, stride, pad, taps_pad);
  DMVAE_CHECK_LAUNCH();
  return 0;
}
extern "C" int dmvae_im2col_nhwc(const void* x, void* col, int n, int h, int w, int c, int ks, int stride, int pad, hipStream_t stream) {
  return dmvae_im2col_nhwc_taps(x, col, n, h, w, c, ks, stride, pad, ks * ks, stream);
}
extern "C" int dmvae_col2im_nhwc(const void* dcol, void* dx, int n, int h, int w, int c, int ks, int stride, int pad, int in_f32,
                                 hipStream_t stream) {
  int ho, wo;
  DMVAE_CHECK_ARG(dcol && dx && n > 0 && h > 0 && w > 0 && c > 0 && c % 8 == 0 && im2col_geom(h, w, ks, stride, pad, &ho, &wo),
                  "col2im_nhwc: bad argument (c must be a multiple of 8; ks 1..7, stride 1..4, pad < ks)");
  const dim3 grid(grid_for((size_t)n * h * w * (c / 8)));
  if (in_f32)
    hipLaunchKernelGGL(col2im_kernel<float>, grid, dim3(256), 0, stream, (const float*)dcol, (bf16*)dx, n, h, w, c, ho, wo, ks, stride, pad);
  else
    hipLaunchKernelGGL(col2im_kernel<bf16>, grid, dim3(256), 0, stream, (const bf16*)dcol, (bf16*)dx, n, h, w, c, ho, wo, ks, stride, pad);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

extern "C" int dmvae_nchw_f32_to_nhwc_bf16(const void* src, void* dst, int n, int c, int hw, int c_pad, hipStream_t stream) {
  DMVAE_CHECK_ARG(src && dst && n > 0 && c > 0 && hw > 0 && c_pad >= c, "nchw_f32_to_nhwc_bf16: bad argument");
  if (c_pad % 8 == 0)
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for((size_t)n * hw * (c_pad / 8))), dim3(256), 0, stream, (const float*)src, (bf16*)dst, n, c, hw, c_pad);
  else
    hipLaunchKernelGGL(nchw_to_nhwc_scalar_kernel, dim3(grid_for((size_t)n * hw)), dim3(256), 0, stream, (const float*)src, (bf16*)dst, n, c, hw, c_pad);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

extern "C" int dmvae_nhwc_to_nchw_f32(const void* src, void* dst, int n, int c, int hw, int c_pad, int src_f32, hipStream_t stream) {
  DMVAE_CHECK_ARG(src && dst && n > 0 && c > 0 && hw > 0 && c_pad >= c, "nhwc_to_nchw_f32: bad argument");
  if (src_f32)
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(grid_for((size_t)n * hw)), dim3(256), 0, stream, (const float*)src, (float*)dst, n, c, hw, c_pad);
  else
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16>, dim3(grid_for((size_t)n * hw)), dim3(256), 0, stream, (const bf16*)src, (float*)dst, n, c, hw, c_pad);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

extern "C" int dmvae_silu_fwd(const void* x, void* y, size_t n, hipStream_t stream) {
  DMVAE_CHECK_ARG(x && y && n % 8 == 0, "silu_fwd: element count must be a multiple of 8");
  if (n == 0) return 0;
  hipLaunchKernelGGL(silu_fwd_kernel, dim3(grid_for(n / 8)), dim3(256), 0, stream, (const bf16*)x, (bf16*)y, n / 8);
  DMVAE_CHECK_LAUNCH();
  return 0;
}
extern "C" int dmvae_silu_bwd(const void* x, const void* dy, void* dx, size_t n, hipStream_t stream) {
  DMVAE_CHECK_ARG(x && dy && dx && n % 8 == 0, "silu_bwd: element count must be a multiple of 8");
  if (n == 0) return 0;
  hipLaunchKernelGGL(silu_bwd_kernel, dim3(grid_for(n / 8)), dim3(256), 0, stream, (const bf16*)x, (const bf16*)dy, (bf16*)dx, n / 8);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

extern "C" int dmvae_softmax_rows_fwd(const void* s, void* p, int rows, int cols, float scale, hipStream_t stream) {
  DMVAE_CHECK_ARG(s && p && rows > 0 && cols > 0, "softmax_rows_fwd: bad argument");
  const dim3 grid((rows + 3) / 4);
  if (cols == 1024) hipLaunchKernelGGL(softmax_fwd_rows_kernel<4>, grid, dim3(256), 0, stream, (const float*)s, (bf16*)p, rows, scale);
  else if (cols == 512) hipLaunchKernelGGL(softmax_fwd_rows_kernel<2>, grid, dim3(256), 0, stream, (const float*)s, (bf16*)p, rows, scale);
  else if (cols == 256) hipLaunchKernelGGL(softmax_fwd_rows_kernel<1>, grid, dim3(256), 0, stream, (const float*)s, (bf16*)p, rows, scale);
  else if (cols == 2048) hipLaunchKernelGGL(softmax_fwd_rows_kernel<8>, grid, dim3(256), 0, stream, (const float*)s, (bf16*)p, rows, scale);
  else hipLaunchKernelGGL(softmax_fwd_kernel, grid, dim3(256), 0, stream, (const float*)s, (bf16*)p, rows, cols, scale);
  DMVAE_CHECK_LAUNCH();
  return 0;
}
extern "C" int dmvae_softmax_rows_bwd(const void* dp, const void* p, void* ds, int rows, int cols, float scale, hipStream_t stream) {
  DMVAE_CHECK_ARG(dp && p && ds && rows > 0 && cols > 0, "softmax_rows_bwd: bad argument");
  const dim3 grid((rows + 3) / 4);
  if (cols == 1024) hipLaunchKernelGGL(softmax_bwd_rows_kernel<4>, grid, dim3(256), 0, stream, (const float*)dp, (const bf16*)p, (bf16*)ds, rows, scale);
  else if (cols == 512) hipLaunchKernelGGL(softmax_bwd_rows_kernel<2>, grid, dim3(256), 0, stream, (const float*)dp, (const bf16*)p, (bf16*)ds, rows, scale);
  else if (cols == 256) hipLaunchKernelGGL(softmax_bwd_rows_kernel<1>, grid, dim3(256), 0, stream, (const float*)dp, (const bf16*)p, (bf16*)ds, rows, scale);
  else if (cols == 2048) hipLaunchKernelGGL(softmax_bwd_rows_kernel<8>, grid, dim3(256), 0, stream, (const float*)dp, (const bf16*)p, (bf16*)ds, rows, scale);
  else hipLaunchKernelGGL(softmax_bwd_kernel, grid, dim3(256), 0, stream, (const float*)dp, (const bf16*)p, (bf16*)ds, rows, cols, scale);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

extern "C" int dmvae_transpose_bf16(const void* src, void* dst, int batch, int rows, int cols, hipStream_t stream) {
  DMVAE_CHECK_ARG(src && dst && batch > 0 && batch < 65536 && rows > 0 && cols > 0, "transpose_bf16: bad argument");
  hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32, batch), dim3(32, 8), 0, stream, (const bf16*)src, (bf16*)dst, rows, cols);
  DMVAE_CHECK_LAUNCH();
  return 0;
}
