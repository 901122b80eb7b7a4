// DiffAug (utils/diffaug.py:43-114; StyleGAN-T's differentiable augmentation) on NCHW f32 images, forward and backward.
//
// The reference composes ~25 ATen kernels per call (pad, meshgrid gather, three broadcast means, a scatter into a mask ...) on the
// [2B,3,256,256] discriminator input; here the chain is one per-image reduction plus one elementwise pass in each direction:
//
//   u = translate(x)         out[h,w] = x[h+th, w+tw], zero fill; th, tw = floor(r0|r1 * (2*delta+1)) - delta
//   v = u + (r2 - 0.5)                                                   brightness
//   s = (v - mean_c v) * 2*r3 + mean_c v                                  saturation (per-pixel channel mean)
//   z = (s - mean_chw s) * (r4 + 0.5) + mean_chw s                        contrast   (per-image mean; = mean_chw u + r2 - 0.5)
//   y = z * mask             mask = 0 on rows/cols clamp(o - c/2 + [0,c)), o = floor(r5|r6 * (size + 1 - c%2))
// with delta = round(size/8) and c = round(size*cutout) evaluated by the caller (Python's round(), as in the reference).
//
// The random draws r[7][B] stay on the device (no host round trip); which of the three stages run is the caller's `flags`
// (bit 0 translate, bit 1 colour, bit 2 cut-out: the outcome of torch.rand(3) <= prob, diffaug.py:66).  Index arithmetic is
// exact (translation / cut-out results are bit-identical to the reference); the colour path differs by f32 summation order only.
// The warm-up blur (:47-63) is not built: every reference call site passes schedule 0.
#include "common.h"
#include "dmvae_hip.h"

namespace dmvae_diffaug {

constexpr int MAXC = 8;

struct Geo {
  int B, C, H, W, dh, dw, ch, cw, flags;
};

__device__ __forceinline__ void draws(const float* __restrict__ r, int b, const Geo& g, int& th, int& tw, int& lo_h, int& hi_h, int& lo_w,
                                      int& hi_w) {
  th = tw = 0;
  if (g.flags & 1) {
    th = (int)floorf(r[0 * g.B + b] * (float)(2 * g.dh + 1)) - g.dh;
    tw = (int)floorf(r[1 * g.B + b] * (float)(2 * g.dw + 1)) - g.dw;
  }
  lo_h = lo_w = 1; hi_h = hi_w = 0;  // empty
  if (g.flags & 4) {
    const int oh = (int)floorf(r[5 * g.B + b] * (float)(g.H + (1 - g.ch % 2)));
    const int ow = (int)floorf(r[6 * g.B + b] * (float)(g.W + (1 - g.cw % 2)));
    if (g.ch > 0 && g.cw > 0) {
      lo_h = min(max(oh - g.ch / 2, 0), g.H - 1); hi_h = min(max(oh - g.ch / 2 + g.ch - 1, 0), g.H - 1);
      lo_w = min(max(ow - g.cw / 2, 0), g.W - 1); hi_w = min(max(ow - g.cw / 2 + g.cw - 1, 0), g.W - 1);
    }
  }
}

// RED_CH blocks per image (row chunks), fixed-order two-stage sum in f64.  mode 0: sum of the translated image (the in-range window of x); mode 1: sum of dy * mask.
// (One block per image read 786 KB with 64 of 256 CUs busy: 125 us per call at 2B = 64 images.)
constexpr int RED_CH = 16;
__global__ __launch_bounds__(256) void reduce_kernel(const float* __restrict__ src, const float* __restrict__ r, double* __restrict__ part,
                                                     Geo g, int mode) {
  const int b = blockIdx.y;
  int th, tw, lo_h, hi_h, lo_w, hi_w;
  draws(r, b, g, th, tw, lo_h, hi_h, lo_w, hi_w);
  const int rows = g.C * g.H;                                   // (channel, row) pairs of the image, W contiguous floats each
  const int rpc = (rows + RED_CH - 1) / RED_CH;
  const int r0 = blockIdx.x * rpc, r1 = min(rows, r0 + rpc);
  const float* img = src + (size_t)b * rows * g.W;
  double acc = 0.0;
  for (int row = r0 + (threadIdx.x >> 6); row < r1; row += 4) {   // a wave per row
    const int h = row % g.H;
    const bool row_in = mode == 0 ? (unsigned)(h - th) < (unsigned)g.H : true;
    if (!row_in) continue;
    const bool row_masked = mode == 1 && h >= lo_h && h <= hi_h;
    float s = 0.f;
    for (int w = threadIdx.x & 63; w < g.W; w += 64) {
      bool take;
      if (mode == 0) take = (unsigned)(w - tw) < (unsigned)g.W;
      else take = !(row_masked && w >= lo_w && w <= hi_w);
      if (take) s += img[(size_t)row * g.W + w];
    }
    acc += (double)s;
  }
  __shared__ double red[4];
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[(size_t)b * RED_CH + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void reduce_final_kernel(const double* __restrict__ part, float* __restrict__ sums, int B, double inv_chw) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double t = 0.0;
  for (int i = 0; i < RED_CH; i++) t += part[(size_t)b * RED_CH + i];
  sums[b] = (float)(t * inv_chw);
}

__global__ __launch_bounds__(256) void fwd_kernel(const float* __restrict__ x, const float* __restrict__ r, const float* __restrict__ mean_u,
                                                  float* __restrict__ y, Geo g) {
  const size_t hw = (size_t)g.H * g.W, total = (size_t)g.B * hw;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int w = i % g.W, h = (i / g.W) % g.H, b = (int)(i / hw);
    int th, tw, lo_h, hi_h, lo_w, hi_w;
    draws(r, b, g, th, tw, lo_h, hi_h, lo_w, hi_w);
    const int sh = h + th, sw = w + tw;
    const bool inside = (unsigned)sh < (unsigned)g.H && (unsigned)sw < (unsigned)g.W;
    float v[MAXC];
    for (int c = 0; c < g.C; c++) v[c] = inside ? x[((size_t)b * g.C + c) * hw + (size_t)sh * g.W + sw] : 0.f;
    if (g.flags & 2) {
      const float br = r[2 * g.B + b] - 0.5f, s1 = r[3 * g.B + b] * 2.f, s2 = r[4 * g.B + b] + 0.5f;
      float m = 0.f;
      for (int c = 0; c < g.C; c++) { v[c] += br; m += v[c]; }
      m /= (float)g.C;
      const float mi = mean_u[b] + br;
      for (int c = 0; c < g.C; c++) {
        const float s = (v[c] - m) * s1 + m;
        v[c] = (s - mi) * s2 + mi;
      }
    }
    const bool cut = h >= lo_h && h <= hi_h && w >= lo_w && w <= hi_w;
    for (int c = 0; c < g.C; c++) y[((size_t)b * g.C + c) * hw + (size_t)h * g.W + w] = cut ? v[c] * 0.f : v[c];
  }
}

// dx[h', w'] = du[h' - th, w' - tw] (zero outside), du = s1*dw + (1 - s1)*mean_c dw, dw = s2*dz + (1 - s2)*mean_chw dz, dz = dy*mask
__global__ __launch_bounds__(256) void bwd_kernel(const float* __restrict__ dy, const float* __restrict__ r, const float* __restrict__ mean_dz,
                                                  float* __restrict__ dx, Geo g) {
  const size_t hw = (size_t)g.H * g.W, total = (size_t)g.B * hw;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int w = i % g.W, h = (i / g.W) % g.H, b = (int)(i / hw);   // position in x
    int th, tw, lo_h, hi_h, lo_w, hi_w;
    draws(r, b, g, th, tw, lo_h, hi_h, lo_w, hi_w);
    const int oh = h - th, ow = w - tw;                               // the output pixel this input pixel was moved to
    const bool inside = (unsigned)oh < (unsigned)g.H && (unsigned)ow < (unsigned)g.W;
    float d[MAXC];
    if (inside) {
      const bool cut = oh >= lo_h && oh <= hi_h && ow >= lo_w && ow <= hi_w;
      for (int c = 0; c < g.C; c++) d[c] = cut ? 0.f : dy[((size_t)b * g.C + c) * hw + (size_t)oh * g.W + ow];
      if (g.flags & 2) {
        const float s1 = r[3 * g.B + b] * 2.f, s2 = r[4 * g.B + b] + 0.5f;
        const float mz = mean_dz[b];
        float m = 0.f;
        for (int c = 0; c < g.C; c++) { d[c] = s2 * d[c] + (1.f - s2) * mz; m += d[c]; }
        m /= (float)g.C;
        for (int c = 0; c < g.C; c++) d[c] = s1 * d[c] + (1.f - s1) * m;
      }
    } else {
      for (int c = 0; c < g.C; c++) d[c] = 0.f;
    }
    for (int c = 0; c < g.C; c++) dx[((size_t)b * g.C + c) * hw + (size_t)h * g.W + w] = d[c];
  }
}

static inline int grid_for(size_t n, int block = 256, int cap = 4096) {
  size_t g = (n + block - 1) / block;
  return (int)(g > (size_t)cap ? cap : (g < 1 ? 1 : g));
}

static int make_geo(Geo& g, int b, int c, int h, int w, int flags, int delta_h, int delta_w, int cut_h, int cut_w) {
  if (b <= 0 || c <= 0 || c > MAXC || h <= 0 || w <= 0 || flags < 0 || flags > 7) return -1;
  if (delta_h < 0 || delta_h > h || delta_w < 0 || delta_w > w || cut_h < 0 || cut_h > h || cut_w < 0 || cut_w > w) return -1;
  g.B = b; g.C = c; g.H = h; g.W = w; g.flags = flags;
  g.dh = delta_h; g.dw = delta_w; g.ch = cut_h; g.cw = cut_w;
  return 0;
}

}  // namespace dmvae_diffaug
using namespace dmvae_diffaug;

extern "C" int dmvae_diffaug_fwd(const void* x, const void* rand01, void* y, void* workspace, int b, int c, int h, int w, int flags,
                                 int delta_h, int delta_w, int cut_h, int cut_w, hipStream_t stream) {
  Geo g;
  DMVAE_CHECK_ARG(x && rand01 && y && workspace, "diffaug_fwd: null pointer");
  DMVAE_CHECK_ARG(make_geo(g, b, c, h, w, flags, delta_h, delta_w, cut_h, cut_w) == 0, "diffaug_fwd: bad argument (channels <= 8, flags 0..7, sizes within the image)");
  if (flags & 2) {
    double* part = reinterpret_cast<double*>((float*)workspace + 2 * ((b + 1) / 2));      // 8-byte aligned behind the b means
    hipLaunchKernelGGL(reduce_kernel, dim3(RED_CH, b), dim3(256), 0, stream, (const float*)x, (const float*)rand01, part, g, 0);
    DMVAE_CHECK_LAUNCH();
    hipLaunchKernelGGL(reduce_final_kernel, dim3((b + 63) / 64), dim3(64), 0, stream, (const double*)part, (float*)workspace, b, 1.0 / ((double)c * h * w));
    DMVAE_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(fwd_kernel, dim3(grid_for((size_t)b * h * w)), dim3(256), 0, stream, (const float*)x, (const float*)rand01,
                     (const float*)workspace, (float*)y, g);
  DMVAE_CHECK_LAUNCH();
  return 0;
}

extern "C" int dmvae_diffaug_bwd(const void* dy, const void* rand01, void* dx, void* workspace, int b, int c, int h, int w, int flags,
                                 int delta_h, int delta_w, int cut_h, int cut_w, hipStream_t stream) {
  Geo g;
  DMVAE_CHECK_ARG(dy && rand01 && dx && workspace, "diffaug_bwd: null pointer");
  DMVAE_CHECK_ARG(make_geo(g, b, c, h, w, flags, delta_h, delta_w, cut_h, cut_w) == 0, "diffaug_bwd: bad argument (channels <= 8, flags 0..7, sizes within the image)");
  if (flags & 2) {
    double* part = reinterpret_cast<double*>((float*)workspace + 2 * ((b + 1) / 2));      // 8-byte aligned behind the b means
    hipLaunchKernelGGL(reduce_kernel, dim3(RED_CH, b), dim3(256), 0, stream, (const float*)dy, (const float*)rand01, part, g, 1);
    DMVAE_CHECK_LAUNCH();
    hipLaunchKernelGGL(reduce_final_kernel, dim3((b + 63) / 64), dim3(64), 0, stream, (const double*)part, (float*)workspace, b, 1.0 / ((double)c * h * w));
    DMVAE_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(bwd_kernel, dim3(grid_for((size_t)b * h * w)), dim3(256), 0, stream, (const float*)dy, (const float*)rand01,
                     (const float*)workspace, (float*)dx, g);
  DMVAE_CHECK_LAUNCH();
  return 0;
}
