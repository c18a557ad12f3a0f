// Probe: how long does a chained launch's bounded wait take when the awaited granules never arrive?  Every workgroup of a
// grid shaped like decode_chain_kernel's (860 x 256 threads at Qwen2-VL-7B shapes) runs the product's poll loop body - one
// relaxed agent-scope (sc1) 8-byte load per thread, the tag compare, gr_poll_abort every 32nd poll, s_sleep 2 - for
// GV_CHAIN_SPIN_MAX polls on granules whose tag never matches.  Prints the wall time of the launch = the bound a stranded
// chained launch costs before it raises the status word.  Build (on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 -I vision-inspection-system_amd/csrc -o tools/probes/poll_period tools/probes/poll_period.hip
#include "decode_common.hip.h"
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void poll_kernel(const gran_t* g, int* status, int* gave_up, int polls) {
  const gran_t* mine = g + (size_t)blockIdx.x * 256 + threadIdx.x;
  bool ok = false;
  for (int it = 0; it < polls && !ok; ++it) {
    if (gr_poll_abort(status, it)) break;
    ok = gr_ok(gr_ld(mine), 7u);
    if (!ok) __builtin_amdgcn_s_sleep(2);
  }
  if (!ok && threadIdx.x == 0) atomicAdd(gave_up, 1);
}

int main(int argc, char** argv) {
  const int grid = argc > 1 ? atoi(argv[1]) : 860;
  gran_t* g; int* status; int* gave_up;
  (void)hipMalloc(&g, (size_t)grid * 256 * 8); (void)hipMalloc(&status, 4); (void)hipMalloc(&gave_up, 4);
  (void)hipMemset(g, 0, (size_t)grid * 256 * 8); (void)hipMemset(status, 0, 4); (void)hipMemset(gave_up, 0, 4);
  hipEvent_t s, e; (void)hipEventCreate(&s); (void)hipEventCreate(&e);
  for (int polls : {256, GV_CHAIN_SPIN_MAX, GV_CHAIN_SPIN_MAX}) {
    (void)hipEventRecord(s);
    poll_kernel<<<grid, 256>>>(g, status, gave_up, polls);
    (void)hipEventRecord(e); (void)hipEventSynchronize(e);
    float ms; (void)hipEventElapsedTime(&ms, s, e);
    printf("%d workgroups x 256 threads, %d polls each on granules that never arrive: %.3f ms = %.2f us per poll\n", grid, polls, ms,
           ms * 1e3 / polls);
  }
  // the same wait when another workgroup has already given up: ends at the next 32nd poll
  int one = 1; (void)hipMemcpy(status, &one, 4, hipMemcpyHostToDevice);
  (void)hipEventRecord(s);
  poll_kernel<<<grid, 256>>>(g, status, gave_up, GV_CHAIN_SPIN_MAX);
  (void)hipEventRecord(e); (void)hipEventSynchronize(e);
  float ms; (void)hipEventElapsedTime(&ms, s, e);
  printf("status word already raised: %.1f us for the whole grid\n", ms * 1e3);
  return 0;
}
