// Probe: how fast can ONE workgroup per CU (8 waves) pull data into LDS with global_load_lds_dwordx4 when nothing else
// runs?  Each workgroup streams "tiles" of 64 KiB (the per-K-tile staging of the 256x256 GEMM: 8 instructions per wave,
// rows of 128 B, 8 rows per instruction) from a region that is (a) L2-resident (2 MiB per workgroup-group, re-read) or
// (b) far larger than the caches.  Counted vmcnt so that 32 KiB (or 48 / 16) stay in flight.  Prints bytes per clock per
// CU at the measured wall time and the nominal 2.4 GHz.  Build: hipcc --offload-arch=gfx950 -O3 -o glds_rate glds_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int MODE>   // 0: all 8 waves issue; 1: only 4 waves issue (the other 4 idle)
__global__ __launch_bounds__(512, 2) void glds_stream(const char* src, size_t region, int tiles, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const size_t wg_base = ((size_t)blockIdx.x * 65536 * 7) % region;
  const int row = lane >> 3, ch = (lane & 7) ^ (row & 7);
  for (int t = 0; t < tiles; ++t) {
    const char* tile = src + (wg_base + (size_t)t * 65536) % region;
    char* dst = lds + (t & 1) * 65536 + wave * 1024;
    if (MODE == 0 || wave < 4) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const char* g = tile + (size_t)(i * 8 + wave) * 1024 + row * 128 + ch * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(dst + i * 8192), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // one tile (8 instructions per wave) stays in flight
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) sink[blockIdx.x] = *(float*)(lds + 128);
}

int main(int argc, char** argv) {
  const int tiles = argc > 1 ? atoi(argv[1]) : 2000;
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  const size_t big = (size_t)4 << 30;
  char* src; float* sink;
  hipMalloc(&src, big); hipMemset(src, 1, big); hipMalloc(&sink, cus * 4);
  hipFuncSetAttribute((const void*)glds_stream<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipFuncSetAttribute((const void*)glds_stream<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  struct { const char* name; size_t region; int mode; } cases[] = {
      {"L2-resident (16 MiB region), 8 waves", (size_t)16 << 20, 0},
      {"L2-resident (16 MiB region), 4 waves", (size_t)16 << 20, 1},
      {"Infinity-Cache-resident (128 MiB), 8 waves", (size_t)128 << 20, 0},
      {"HBM (4 GiB region), 8 waves", big, 0}};
  for (auto& c : cases) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(s);
      if (c.mode == 0) glds_stream<0><<<cus, 512, 131072>>>(src, c.region, tiles, sink);
      else glds_stream<1><<<cus, 512, 131072>>>(src, c.region, tiles, sink);
      hipEventRecord(e); hipEventSynchronize(e);
      float ms; hipEventElapsedTime(&ms, s, e);
      if (rep == 0) continue;
      const double bytes = (double)tiles * 65536 * (c.mode == 0 ? 1.0 : 0.5);
      const double per_cu = bytes / (ms * 1e-3);
      printf("%-46s %7.3f ms  %6.1f GB/s per CU  %5.1f B/clk/CU at 2.4 GHz  chip %5.2f TB/s\n", c.name, ms, per_cu / 1e9,
             per_cu / 2.4e9, per_cu * cus / 1e12);
    }
  }
  if (hipGetLastError() != hipSuccess) { printf("HIP error\n"); return 1; }
  return 0;
}
