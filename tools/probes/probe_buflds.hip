// Probe: buffer_load_dwordx4 ... offen lds (LDS-DMA through a buffer descriptor) on gfx950.
//  (a) out-of-range voffset -> zeros land in LDS?   (b) soffset added to the address?   (c) descriptor base below the
//  allocation + positive offsets reaches the right bytes?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const char* p, unsigned nbytes, unsigned* out, unsigned soff, const unsigned* voffs, long long shift) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  for (int i = threadIdx.x; i < 256; i += 64) ((unsigned*)smem)[i] = 0xdeadbeefu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(p - shift), 0, nbytes + (unsigned)shift, 0x00020000);
  const unsigned vo = voffs[threadIdx.x];
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)smem, 16, vo, soff, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = ((unsigned*)smem)[i];
}
int main() {
  const unsigned N = 1 << 16;
  std::vector<unsigned> h(N / 4);
  for (unsigned i = 0; i < N / 4; i++) h[i] = i;   // word i holds i
  char* d; unsigned *dout, *dvo;
  hipMalloc(&d, N); hipMalloc(&dout, 1024); hipMalloc(&dvo, 256);
  hipMemcpy(d, h.data(), N, hipMemcpyHostToDevice);
  for (int test = 0; test < 3; test++) {
    std::vector<unsigned> vo(64), o(256);
    long long shift = test == 2 ? 4096 : 0;
    unsigned soff = test >= 1 ? 512 : 0;
    for (int l = 0; l < 64; l++) vo[l] = (l % 4 == 3) ? 0x80000000u : (unsigned)(l * 32 + shift);   // every 4th lane out of range
    hipMemcpy(dvo, vo.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, d, N, dout, soff, dvo, shift);
    hipMemcpy(o.data(), dout, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++)
      for (int e = 0; e < 4; e++) {
        unsigned want = (l % 4 == 3) ? 0u : (unsigned)((l * 32 + soff) / 4 + e);
        if (o[l * 4 + e] != want) { if (bad < 4) printf("  test %d lane %d e %d got %08x want %08x\n", test, l, e, o[l * 4 + e], want); bad++; }
      }
    printf("BUFLDS test %d (soff=%u shift=%lld): mismatches %d\n", test, soff, shift, bad);
  }
  return 0;
}
