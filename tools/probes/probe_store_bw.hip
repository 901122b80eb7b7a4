// Probe: what bounds conv_pp's epilogue stores -- the CU, the XCD or the chip?
// Every active workgroup (512 threads, 128 KB of LDS so that it owns its CU) writes `reps` tiles of 128 KB the way the epilogue does: a wave-instruction
// is global_store_dwordx4 over 8 pixel rows x 128 contiguous bytes (rows `stride` bytes apart), 16 instructions per wave and tile.  Timed with s_memtime per
// block (shader cycles) and hipEvents (wall).  Active sets: N blocks spread over all XCDs (block b -> XCD b % 8), or only the blocks of XCD 0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct A { char* out; int reps, nt, stride, xcd_only, gap, seg64; unsigned long long* st; };
__global__ __launch_bounds__(512) void k(A a) {
  extern __shared__ char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (a.xcd_only >= 0 && (int)(blockIdx.x & 7) != a.xcd_only) return;
  smem[tid] = 0;
  const unsigned active_idx = a.xcd_only >= 0 ? blockIdx.x >> 3 : blockIdx.x;
  f32x4 v = {(float)tid, 1.f, 2.f, 3.f};
  const int cl = lane & 7, rg = lane >> 3;
  __syncthreads();
  unsigned long long burst = 0, real = 0;
  // tile = 256 rows; wave w: rows (w & 3) * 64 .. +63, column half (w >> 2) -> 128 B of the row's 256 B (2 cout halves)
  for (int r = 0; r < a.reps; r++) {
    const unsigned long long b0 = __builtin_amdgcn_s_memtime(), r0 = wall_clock64();
    char* tile = a.out + ((size_t)active_idx * a.reps + r) * (size_t)(256 * a.stride);
    if (a.seg64) {  // the MFMA result layout with permuted weight rows: 16 pixel rows x 64 contiguous bytes per instruction, 16 instructions per wave
#pragma unroll
      for (int ip = 0; ip < 4; ip++)
#pragma unroll
        for (int jb = 0; jb < 4; jb++) {
          const int row = (wave & 3) * 64 + jb * 16 + (lane & 15);
          char* p = tile + (size_t)row * a.stride + (wave >> 2) * 256 + ip * 64 + (lane >> 4) * 16;
          if (a.nt) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
          else *reinterpret_cast<f32x4*>(p) = v;
        }
    } else if (a.seg64 == 2) {  // 4 pixel rows x 256 contiguous bytes per instruction (all 128 couts of the wave per row)
#pragma unroll
      for (int it = 0; it < 16; it++) {
        const int row = (wave & 3) * 64 + it * 4 + (lane >> 4);
        char* p = tile + (size_t)row * a.stride + (wave >> 2) * 256 + (lane & 15) * 16;
        if (a.nt) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
        else *reinterpret_cast<f32x4*>(p) = v;
      }
    } else {
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int it = 0; it < 8; it++) {
        // j: which 128-B column block of the wave's 256 B (emulates two staging passes); 8 instr x 8 rows = 64 rows
        const int row = (wave & 3) * 64 + it * 8 + rg;
        char* p = tile + (size_t)row * a.stride + ((wave >> 2) * 2 + j) * 128 + cl * 16;
        if (a.nt) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
        else *reinterpret_cast<f32x4*>(p) = v;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    burst += __builtin_amdgcn_s_memtime() - b0;
    real += wall_clock64() - r0;
    for (int i = 0; i < a.gap; i++) __builtin_amdgcn_s_sleep(64);  // idle time between bursts (a main loop's worth), so that the memory system drains
  }
  if (tid == 0) { a.st[blockIdx.x * 2] = real; a.st[blockIdx.x * 2 + 1] = burst; }
}
int main(int argc, char** argv) {
  const int reps = 16;
  const size_t bytes = (size_t)256 * reps * 256 * 1024;  // worst case stride 1024
  char* d; unsigned long long* st;
  hipMalloc(&d, bytes); hipMalloc(&st, 256 * 16);
  hipMemset(d, 0, bytes);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("tile = 128 KB (256 rows x 512 B of a row); %d tiles per block; cycles = s_memtime per tile, mean over active blocks\n", reps);
  for (int stride : {512})
    for (int nt : {1})
      for (int gap : {0, 40})
      for (int seg64 : {0, 2}) {
        struct Cfg { int grid, xcd; const char* name; };
        const Cfg cfgs[] = {{8, -1, "1 CU per XCD (8)"}, {256, 0, "32 CUs, XCD 0 only"}, {64, -1, "8 CUs per XCD (64)"}, {128, -1, "16 per XCD (128)"}, {256, -1, "all 256 CUs"}};
        for (const Cfg& c : cfgs) {
          A a{d, reps, nt, stride, c.xcd, gap, seg64, st};
          for (int w = 0; w < 2; w++) {
            hipMemset(st, 0, 256 * 16);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(c.grid), dim3(512), 128 * 1024, 0, a);
            hipEventRecord(e1);
            hipDeviceSynchronize();
          }
          float ms; hipEventElapsedTime(&ms, e0, e1);
          std::vector<unsigned long long> h(512);
          hipMemcpy(h.data(), st, 512 * 8, hipMemcpyDeviceToHost);
          double sum = 0, rsum = 0; int n = 0;
          for (int b = 0; b < c.grid; b++) if (h[2 * b + 1]) { sum += (double)h[2 * b + 1]; rsum += (double)h[2 * b]; n++; }
          const double cyc = sum / n / reps, ns = rsum / n / reps * 10.0;  // wall_clock64: 100 MHz
          printf("stride %4d %s gap %2d %s | %-22s: active %3d, %8.0f cycles per 128-KB burst, %6.1f B/clk/CU, %6.0f ns = %5.1f GB/s per CU, %6.2f TB/s over the active CUs, launch %7.1f us\n", stride, nt ? "nt   " : "plain", gap, seg64 == 2 ? "4x256B" : (seg64 ? "16x64B " : "8x128B"),
                 c.name, n, cyc, 131072.0 / cyc, ns, 131072.0 / ns, 131072.0 / ns * n / 1e3, ms * 1e3);
        }
      }
  return 0;
}
