// Probe: which lane's scale operand of v_mfma_scale_f32_16x16x128_f8f6f4 applies to which (row / column, 32-wide K block)?
// A = B = all ones (e4m3 1.0), so D[i][j] = sum over the four K blocks of 32 * sA(i, blk) * sB(j, blk).  One lane at a time
// gets scale 2.0 (E8M0 128), everything else 1.0, with the operand data of ONE K block live (the others zero): the column
// (row) whose result doubles names the (index, block) that lane's scale governs.  Also: which byte OPSEL picks.
// The instruction's K order (found with the first version of this probe, which assumed a lane's 8 VGPRs are 32 consecutive
// K elements and saw every scale govern HALF of two lanes' data): lane (l15, h) holds K elements 16 h .. 16 h + 15 in its
// first four VGPRs and 64 + 16 h .. + 15 in its last four; K block s = elements 32 s .. 32 s + 31 = VGPR half (s >> 1) of
// the lanes h = 2 (s & 1) and 2 (s & 1) + 1.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma_scale_probe tools/probes/mfma_scale_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int OPSEL>
__global__ void probe(float* C, int hot_lane, int live_blk, int which, unsigned hot_word, unsigned cold_word) {
  const int lane = threadIdx.x, h = lane >> 4;
  // e4m3 1.0 in the bytes of K block live_blk, zeros elsewhere
  const int lo = ((h >> 1) == (live_blk & 1) && (live_blk >> 1) == 0) ? 0x38383838 : 0;
  const int hi = ((h >> 1) == (live_blk & 1) && (live_blk >> 1) == 1) ? 0x38383838 : 0;
  const i32x8 a = {lo, lo, lo, lo, hi, hi, hi, hi}, b = a;
  const int hot = (int)hot_word, cold = (int)cold_word;
  const int sa = (which == 0 && lane == hot_lane) ? hot : cold;
  const int sb = (which == 1 && lane == hot_lane) ? hot : cold;
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, OPSEL, sa, OPSEL, sb);
#pragma unroll
  for (int r = 0; r < 4; ++r) C[(4 * h + r) * 16 + (lane & 15)] = c[r];   // D[row = 4 h + r][col = lane & 15]
}

int main() {
  float* dC; hipMalloc(&dC, 256 * 4);
  float hC[256];
  for (int which = 0; which < 2; ++which) {
    printf("=== scale of operand %s (first / second builtin operand), byte 0, OPSEL 0 ===\n", which == 0 ? "A" : "B");
    for (int L = 0; L < 64; ++L) {
      for (int blk = 0; blk < 4; ++blk) {
        probe<0><<<1, 64>>>(dC, L, blk, which, 0x7f7f7f80u, 0x7f7f7f7fu);
        hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
        // rows / columns whose value is 64 instead of 32
        int nr = 0, nc = 0, r0 = -1, c0 = -1;
        for (int i = 0; i < 16; ++i) { int all = 1; for (int j = 0; j < 16; ++j) all &= (hC[i * 16 + j] == 64.f); if (all) { ++nr; r0 = i; } }
        for (int j = 0; j < 16; ++j) { int all = 1; for (int i = 0; i < 16; ++i) all &= (hC[i * 16 + j] == 64.f); if (all) { ++nc; c0 = j; } }
        if (nr || nc) printf("lane %2d (l15 %2d, h %d): doubles %s %2d for K block %d\n", L, L & 15, L >> 4, nr ? "D row" : "D col", nr ? r0 : c0, blk);
      }
    }
  }
  printf("=== OPSEL: scale word 0x83828180 (bytes 3..0 = 2^4, 2^3, 2^2, 2^1) on lane 0 as the B scale, K block 0 live ===\n");
  probe<0><<<1, 64>>>(dC, 0, 0, 1, 0x83828180u, 0x7f7f7f7fu); hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
  printf("OPSEL 0: D[0][0..3] = %g %g %g %g (32 x scale of the affected entry)\n", hC[0], hC[1], hC[2], hC[3]);
  probe<1><<<1, 64>>>(dC, 0, 0, 1, 0x83828180u, 0x7f7f7f7fu); hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
  printf("OPSEL 1: D[0][0..3] = %g %g %g %g\n", hC[0], hC[1], hC[2], hC[3]);
  probe<2><<<1, 64>>>(dC, 0, 0, 1, 0x83828180u, 0x7f7f7f7fu); hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
  printf("OPSEL 2: D[0][0..3] = %g %g %g %g\n", hC[0], hC[1], hC[2], hC[3]);
  probe<3><<<1, 64>>>(dC, 0, 0, 1, 0x83828180u, 0x7f7f7f7fu); hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
  printf("OPSEL 3: D[0][0..3] = %g %g %g %g\n", hC[0], hC[1], hC[2], hC[3]);
  printf("=== cold word 0x7f in byte 0 only (0x0000007f), OPSEL 0, K block 0 live: expect 32 everywhere ===\n");
  probe<0><<<1, 64>>>(dC, -1, 0, 1, 0, 0x0000007fu); hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
  printf("D[0][0] = %g, D[5][9] = %g\n", hC[0], hC[5 * 16 + 9]);
  return 0;
}
