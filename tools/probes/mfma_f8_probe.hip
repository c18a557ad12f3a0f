// Probe: operand lane map of v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands (exact small-integer data).
// Hypothesis: lane l holds A[row l&15][k = 32*(l>>4) + j], j = 0..31 (32 bytes = 8 VGPRs), same for B[k][col l&15].
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const uint8_t* A, const uint8_t* B, float* C) {  // A [16][128], B^T [16][128] (row n, k contiguous)
  const int lane = threadIdx.x, l15 = lane & 15, h = lane >> 4;
  i32x8 a, b;
  const int* ap = (const int*)(A + l15 * 128 + h * 32);
  const int* bp = (const int*)(B + l15 * 128 + h * 32);
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = ap[i]; b[i] = bp[i]; }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  // cbsz / blgp = 0 -> fp8 (e4m3) for A and B; scales: E8M0 127 = 2^0 in every byte
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
#pragma unroll
  for (int r = 0; r < 4; ++r) C[(4 * h + r) * 16 + l15] = c[r];   // standard C map: col = lane&15, row = 4*(lane>>4)+r
}

static uint8_t f8(int v) {  // small integers 0..8 as e4m3: exact
  static const uint8_t t[9] = {0x00, 0x38, 0x40, 0x44, 0x48, 0x4a, 0x4c, 0x4e, 0x50};
  return t[v];
}

int main() {
  uint8_t hA[16 * 128], hB[16 * 128];
  int iA[16][128], iB[16][128];
  for (int m = 0; m < 16; ++m) for (int k = 0; k < 128; ++k) { iA[m][k] = (m * 7 + k * 3) % 5; hA[m * 128 + k] = f8(iA[m][k]); }
  for (int n = 0; n < 16; ++n) for (int k = 0; k < 128; ++k) { iB[n][k] = (n * 5 + k * 11 + 1) % 7; hB[n * 128 + k] = f8(iB[n][k]); }
  uint8_t *dA, *dB; float* dC;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, 256 * 4);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(dA, dB, dC);
  float hC[256];
  hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
    int ref = 0;
    for (int k = 0; k < 128; ++k) ref += iA[m][k] * iB[n][k];
    if ((int)hC[m * 16 + n] != ref) { if (bad < 5) printf("C[%d][%d] = %g, expected %d\n", m, n, hC[m * 16 + n], ref); ++bad; }
  }
  printf("mismatches: %d of 256\n", bad);
  return 0;
}
