// Probe: what does a device-wide barrier between the phases of a persistent kernel cost on MI355X, compared with a
// kernel boundary inside a hipGraph?  One workgroup per CU; every barrier = __syncthreads, one agent-scope release +
// fetch_add by thread 0, a BOUNDED spin on the counter (the probe can never hang: a missed barrier sets a flag), an
// acquire, __syncthreads.  Between barriers every workgroup does one dependent global load + store (the shape of a
// small decode kernel).  Compared with the same work as N separate launches in a graph.
// Build: hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target, int* fail) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++spins > (1 << 22)) { *fail = 1; break; }   // bounded: never hangs
      __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void phases_persistent(unsigned* counter, int* fail, float* buf, int nphase, unsigned base) {
  const int n = gridDim.x * 256, i = blockIdx.x * 256 + threadIdx.x;
  for (int p = 0; p < nphase; ++p) {
    const float v = buf[(p & 1) * n + (i * 7 + 13) % n];   // dependent read of the previous phase's output
    buf[((p + 1) & 1) * n + i] = v * 0.999f + 1.0f;
    grid_barrier(counter, base + (unsigned)(p + 1) * gridDim.x, fail);
  }
}

__global__ __launch_bounds__(256) void phase_kernel(float* buf, int p) {
  const int n = gridDim.x * 256, i = blockIdx.x * 256 + threadIdx.x;
  const float v = buf[(p & 1) * n + (i * 7 + 13) % n];
  buf[((p + 1) & 1) * n + i] = v * 0.999f + 1.0f;
}

int main(int argc, char** argv) {
  const int nphase = argc > 1 ? atoi(argv[1]) : 200;
  hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  unsigned* counter; int* fail; float* buf;
  (void)hipMalloc(&counter, 4); (void)hipMalloc(&fail, 4); (void)hipMalloc(&buf, 2 * cus * 256 * 4);
  (void)hipMemset(counter, 0, 4); (void)hipMemset(fail, 0, 4); (void)hipMemset(buf, 0, 2 * cus * 256 * 4);
  hipStream_t st; (void)hipStreamCreate(&st);
  hipEvent_t s, e; (void)hipEventCreate(&s); (void)hipEventCreate(&e);
  float ms;
  unsigned base = 0;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(s, st);
    phases_persistent<<<cus, 256, 0, st>>>(counter, fail, buf, nphase, base);
    (void)hipEventRecord(e, st); (void)hipEventSynchronize(e);
    base += (unsigned)nphase * cus;
    (void)hipEventElapsedTime(&ms, s, e);
    if (rep) printf("persistent kernel, %d phases with grid barriers: %.1f us total, %.2f us per phase\n", nphase, ms * 1e3, ms * 1e3 / nphase);
  }
  int hfail = 0; (void)hipMemcpy(&hfail, fail, 4, hipMemcpyDeviceToHost);
  printf("barrier timeouts: %d\n", hfail);
  // the same phases as separate launches in a graph
  hipGraph_t g; hipGraphExec_t ge;
  (void)hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  for (int p = 0; p < nphase; ++p) phase_kernel<<<cus, 256, 0, st>>>(buf, p);
  (void)hipStreamEndCapture(st, &g);
  (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(s, st);
    (void)hipGraphLaunch(ge, st);
    (void)hipEventRecord(e, st); (void)hipEventSynchronize(e);
    (void)hipEventElapsedTime(&ms, s, e);
    if (rep) printf("hipGraph of %d kernels:                          %.1f us total, %.2f us per kernel\n", nphase, ms * 1e3, ms * 1e3 / nphase);
  }
  return hfail;
}
