// Probe: does the scalar offset take part in a raw buffer descriptor's range check (gfx950)?  The conv epilogue addresses rows of a tile as
// voffset (the lane's first row) + soffset (wave-uniform row step) and relies on rows past the end of the tensor being dropped.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned* buf, unsigned nrec, unsigned soff, unsigned* back) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(buf, 0, nrec, 0x00020000);
  const u32x4 v = {0xAAAA0000u + threadIdx.x, 1, 2, 3};
  __builtin_amdgcn_raw_buffer_store_b128(v, r, threadIdx.x * 16, soff, 0);
  const u32x4 l = __builtin_amdgcn_raw_buffer_load_b128(r, threadIdx.x * 16, soff, 0);
  back[threadIdx.x] = l.x;
}
int main() {
  unsigned *d, *back;
  const unsigned total = 8192;
  hipMalloc(&d, total); hipMalloc(&back, 256);
  for (unsigned soff : {0u, 512u, 1024u, 1536u}) {
    hipMemset(d, 0, total);
    const unsigned nrec = 2048;  // lanes 0..63 x 16 B = 1024 B of voffset range; with soff the accesses reach 1024 + soff
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, nrec, soff, back);
    std::vector<unsigned> h(total / 4), b(64);
    hipMemcpy(h.data(), d, total, hipMemcpyDeviceToHost); hipMemcpy(b.data(), back, 256, hipMemcpyDeviceToHost);
    unsigned last = 0, beyond = 0, ld_nonzero_beyond = 0;
    for (unsigned i = 0; i < total / 4; i++) if (h[i]) { last = i * 4; if (i * 4 >= nrec) beyond++; }
    for (int l = 0; l < 64; l++) if (l * 16 + soff >= nrec && b[l]) ld_nonzero_beyond++;
    printf("num_records %u, soffset %u: accesses span bytes [%u, %u); last byte written %u; dwords written past num_records: %u; loads past the end returning data: %u\n", nrec, soff,
           soff, soff + 1024, last + 4, beyond, ld_nonzero_beyond);
  }
  return 0;
}
