// Probe: semantics of ds_read_b64_tr_b16 and global_load_lds(16B) on gfx950.
// Build: hipcc --offload-arch=gfx950 -O2 probe_tr16.hip -o probe_tr16 ; prints lane->element maps.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define LDSP(p) ((__attribute__((address_space(3))) s16x4*)(p))

__global__ void k_tr(short* out, int mode, int rowstride) {
  __shared__ __attribute__((aligned(16))) short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
  __syncthreads();
  int l = threadIdx.x;
  int off;
  if (mode == 0) off = l * 4;                       // lane-linear
  else {                                            // lane (4r+q) of group g -> row r, cols 16g+4q
    int g = l >> 4, r = (l & 15) >> 2, q = l & 3;
    off = r * rowstride + g * 16 + q * 4;
  }
  s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDSP(lds + off));
  for (int j = 0; j < 4; j++) out[l * 4 + j] = t[j];
}

// glds: each lane copies 16B from global src (per-lane address) to LDS linear base+lane*16
__global__ void k_glds(const short* g, short* out) {
  __shared__ __attribute__((aligned(16))) short lds[2048];
  int l = threadIdx.x;            // 128 threads = 2 waves
  int w = l >> 6;
  // wave w writes lds[w*512 .. w*512+511]; per-lane source = reversed chunk order
  const short* src = g + (w * 64 + (63 - (l & 63))) * 8;
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
      (__attribute__((address_space(3))) void*)(lds + w * 512), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = l; i < 1024; i += 128) out[i] = lds[i];
}

// MFMA 32x32x16 bf16 layout check: D = A*B, A[i][k], B[k][j] with asymmetric B.
__global__ void k_mfma(const __bf16* A, const __bf16* Bt, float* D) {
  int l = threadIdx.x;
  bf16x8 a, b;
  for (int e = 0; e < 8; e++) { a[e] = A[(l & 31) * 16 + (l >> 5) * 8 + e]; b[e] = Bt[(l & 31) * 16 + (l >> 5) * 8 + e]; }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; r++) { int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5); int col = l & 31; D[row * 32 + col] = c[r]; }
}

int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  std::vector<short> h(256);
  for (int mode = 0; mode < 2; mode++) {
    int rs = mode ? 136 : 0;
    hipLaunchKernelGGL(k_tr, 1, 64, 0, 0, d, mode, rs);
    hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
    printf("TR16 mode %d (rowstride %d): lane: e0 e1 e2 e3\n", mode, rs);
    int bad = 0;
    for (int l = 0; l < 64; l++) {
      if (l < 20 || l >= 60) printf("  %2d: %5d %5d %5d %5d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
      for (int j = 0; j < 4; j++) {
        int exp = mode == 0 ? ((l & 15) + j * 16 + (l >> 4) * 64) : (j * rs + (l >> 4) * 16 + (l & 15));
        if (h[l*4+j] != exp) bad++;
      }
    }
    printf("TR16 mode %d hypothesis mismatches: %d\n", mode, bad);
  }
  // glds
  {
    short *g, *o; hipMalloc(&g, 2048 * 2); hipMalloc(&o, 2048 * 2);
    std::vector<short> hg(1024), ho(1024);
    for (int i = 0; i < 1024; i++) hg[i] = (short)i;
    hipMemcpy(g, hg.data(), 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_glds, 1, 128, 0, 0, g, o);
    hipMemcpy(ho.data(), o, 2048, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 1024; i++) { int w = i / 512, c = (i % 512) / 8, e = i % 8; int exp = (w * 64 + (63 - c)) * 8 + e; if (ho[i] != exp) bad++; }
    printf("GLDS16 per-lane-source/linear-dest mismatches: %d (ho[0..3]=%d %d %d %d)\n", bad, ho[0], ho[1], ho[2], ho[3]);
  }
  // mfma
  {
    std::vector<__bf16> A(512), Bt(512); std::vector<float> Af(512), Bf(512), D(1024), R(1024, 0.f);
    for (int i = 0; i < 32; i++) for (int k = 0; k < 16; k++) { float a = (float)((i * 3 + k * 5) % 7 - 3); float b = (float)((i * 2 + k * 7 + (i > 10)) % 5 - 2); Af[i*16+k] = a; Bf[i*16+k] = b; A[i*16+k] = (__bf16)a; Bt[i*16+k] = (__bf16)b; }
    for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) for (int k = 0; k < 16; k++) R[i*32+j] += Af[i*16+k] * Bf[j*16+k];
    __bf16 *dA, *dB; float* dD; hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 4096);
    hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, Bt.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_mfma, 1, 64, 0, 0, dA, dB, dD);
    hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 1024; i++) if (D[i] != R[i]) bad++;
    printf("MFMA32x32x16 layout (A[i][k] rows, B^T[j][k], D[row][col]) mismatches: %d\n", bad);
  }
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("device %s CUs %d clock %d kHz mem %zu GB\n", p.gcnArchName, p.multiProcessorCount, p.clockRate, p.totalGlobalMem >> 30);
  return 0;
}
