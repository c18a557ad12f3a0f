// BASELINE configs[4]: fp8 (OCP e4m3) weights AND activations on the CDNA4 block-scaled MFMA
// (v_mfma_scale_f32_16x16x128_f8f6f4 with unit E8M0 block scales: 2x the bf16 MFMA rate, MI355X_MICROARCH.md).
//
//   C[M, N(/2)] = act((A_q W_q^T) * sa[m] * sw[n] + bias) + R
//   A_q [M][lda] e4m3 bytes with a per-ROW (per-token) f32 scale sa, W_q [N][ldw] e4m3 bytes with a per-output-row
//   f32 scale sw (hip.quantize_fp8_rows), f32 accumulation, bf16 output; same epilogues as vis_gemm_bf16.
//
// Structure = the 256x256 bf16 kernel (gemm_bf16.hip) with the K-step doubled to 128 elements - which is again 128
// BYTES per tile row, so the LDS-DMA staging, the source-side XOR swizzle and the two 64 KiB buffers are identical:
//  * 512 threads = 8 waves (2 x 4), 128 x 64 outputs per wave (8 x 4 MFMA tiles, 128 accumulator VGPRs);
//  * operand lane map (checked with exact integer data, tools/probes/mfma_f8_probe.hip): lane l holds
//    X[row l & 15][k = 32 (l >> 4) + j], j = 0..31 - i.e. 16-byte chunks 2h and 2h+1 of its row (two ds_read_b128
//    at the swizzled positions), 8 VGPRs per tile; C/D as for bf16;
//  * a K-step is 32 MFMAs of K = 128 per wave.  Its operands are fat (96 VGPRs per K-step), so fragments are NOT
//    double-buffered across K-steps; instead the step is split by m: [B, A0] -> 16 MFMAs while A1 is read ->
//    one barrier (all reads of this buffer done, next stage landed) -> refill of this buffer -> 16 MFMAs.
#include "common.hip.h"

#define F8_B 256
// XOR applied to the 16-byte chunk index of tile row r in the LDS image.  A lane of the K = 128 MFMA reads TWO chunks of
// its row (2h and 2h + 1), and `r & 7` - the swizzle of the bf16 kernels, whose lanes read one chunk - put every
// ds_read_b128 lane group ({0-3, 12-15, 20-27}, ...) on 8 of the 16 slots of the 256-byte bank row twice
// (profiles/r04_fp8_gemm_pmc.txt: SQ_LDS_BANK_CONFLICT = 4 x SQ_ACTIVE_INST_LDS on the gate/up shape).  With bits 1 and 3 of
// the row number the two row sets of a group (rows {0-3, 12-15} at chunk c, rows {4-11} at chunk c ^ 2) take disjoint
// halves of the chunk positions: conflict-free for both reads in all four groups (exhaustive check in the r04 notes).
#ifdef F8_SWZ_OLD      // A/B build only (tools/ab_bench.sh): the r03 swizzle
#define F8_SWZ(r) ((r) & 7)
#else
#define F8_SWZ(r) ((((r) >> 1) & 1) | ((((r) >> 3) & 1) << 2))
#endif
#define F8_BK 128                                  // K elements = bytes per tile row per K-step
#define F8_STAGE_BYTES (2 * F8_B * F8_BK)          // 65536
#define F8_SCALE_BYTES 2048                        // ping-pong kernel: the tile's row + column scales
#define F8_LDS_BYTES (2 * F8_STAGE_BYTES + F8_SCALE_BYTES)

#include "gemm_epilogue.hip.h"
enum { F8_ACT_NONE = ACT_NONE, F8_ACT_QUICKGELU = ACT_QUICKGELU, F8_ACT_GELU_ERF = ACT_GELU_ERF, F8_ACT_SWIGLU = ACT_SWIGLU };

typedef int i32x8 __attribute__((ext_vector_type(8)));

struct GemmF8Args {
  const uint8_t* A;
  const uint8_t* W;
  const float* sa;      // [M]
  const float* sw;      // [N]
  const bf16_t* bias;   // [N] or null
  const bf16_t* R;      // [M, ldr] or null
  bf16_t* C;
  int M, N, K, lda, ldw, ldc, ldr, act, tiles_m, tiles_n;
  float* part;   // split-K: f32 partial sums [ksplit][M][N] (unscaled), or null
  int ksplit;
  int wide;      // host-checked preconditions of gemm_epilogue_wide (gemm_epilogue.hip.h) hold
  int nt;        // wide epilogue: non-temporal C stores
};

__device__ __forceinline__ float f8_act(float x, int act) {
  if (act == F8_ACT_QUICKGELU) return quickgelu_fast(x);
  if (act == F8_ACT_GELU_ERF) return gelu_erf(x);
  return x;
}

// 64 x 64 output block of one wave: lane holds D[n = nbase + 16 j + 4 h + r][m = mbase + 16 i + l15] (W was the A operand)
__device__ __forceinline__ void f8_epilogue(const GemmF8Args& p, f32x4 (&acc)[4][4], int mbase, int nbase, int l15, int h) {
  const bool swiglu = (p.act == F8_ACT_SWIGLU);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = mbase + i * 16 + l15;
    if (m >= p.M) continue;
    const float sam = p.sa[m];
    if (swiglu) {
#pragma unroll
      for (int j = 0; j < 4; j += 2) {
        const int n = nbase + j * 16 + 4 * h;  // gate rows; the matching up rows are n + 16
        if (n >= p.N) continue;
        const int oc = (nbase >> 1) + (j >> 1) * 16 + 4 * h;
        const f32x4 sg = *(const f32x4*)(p.sw + n), su = *(const f32x4*)(p.sw + n + 16);
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float g = acc[i][j][r] * sam * sg[r], u = acc[i][j + 1][r] * sam * su[r];
          v[r] = silu_fast(g) * u;
        }
        u32x2 o;
        o[0] = pack2bf(v[0], v[1]);
        o[1] = pack2bf(v[2], v[3]);
        *(u32x2*)(p.C + (size_t)m * p.ldc + oc) = o;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = nbase + j * 16 + 4 * h;
        if (n >= p.N) continue;
        const f32x4 s4 = *(const f32x4*)(p.sw + n);
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] * sam * s4[r];
        if (p.bias) {
          const u32x2 b = *(const u32x2*)(p.bias + n);
          v[0] += __uint_as_float(b[0] << 16);
          v[1] += __uint_as_float(b[0] & 0xffff0000u);
          v[2] += __uint_as_float(b[1] << 16);
          v[3] += __uint_as_float(b[1] & 0xffff0000u);
        }
        if (p.act != F8_ACT_NONE) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = f8_act(v[r], p.act);
        }
        if (p.R) {
          const u32x2 rr = *(const u32x2*)(p.R + (size_t)m * p.ldr + n);
          v[0] += __uint_as_float(rr[0] << 16);
          v[1] += __uint_as_float(rr[0] & 0xffff0000u);
          v[2] += __uint_as_float(rr[1] << 16);
          v[3] += __uint_as_float(rr[1] & 0xffff0000u);
        }
        u32x2 o;
        o[0] = pack2bf(v[0], v[1]);
        o[1] = pack2bf(v[2], v[3]);
        *(u32x2*)(p.C + (size_t)m * p.ldc + n) = o;
      }
    }
  }
}

// Wide form (r05): the accumulators take their two scales - (acc * sa[m]) * sw[n], the direct epilogue's order - on
// their way into the LDS-staged 16-byte-per-lane epilogue of the bf16 kernels (same accumulator layout), which
// applies bias / activation / residual exactly as f8_epilogue does.  tools/gemm_kscan.py, KS_FP8=1, M = 19600 N = 5120: the
// direct form's store tail (8-byte pieces of 16 rows per instruction) cost 14 us per round of 256 tiles against 5.4 us for bf16.
template <int MI>
struct F8Scale {
  float sam[MI];
  f32x4 s4[4];
  __device__ __forceinline__ F8Scale(const GemmF8Args& p, int mbase, int nbase, int l15, int h) {
#pragma unroll
    for (int i = 0; i < MI; ++i) sam[i] = p.sa[min(mbase + i * 16 + l15, p.M - 1)];
#pragma unroll
    for (int j = 0; j < 4; ++j) s4[j] = *(const f32x4*)(p.sw + min(nbase + j * 16 + 4 * h, p.N - 4));
  }
  // from the tile's LDS copy (ping-pong kernel): lsa = the wave's 128 row scales, lsw = its 64 column scales
  __device__ __forceinline__ F8Scale(const float* lsa, const float* lsw, int l15, int h) {
#pragma unroll
    for (int i = 0; i < MI; ++i) sam[i] = lsa[i * 16 + l15];
#pragma unroll
    for (int j = 0; j < 4; ++j) s4[j] = *(const f32x4*)(lsw + j * 16 + 4 * h);
  }
  __device__ __forceinline__ f32x4 operator()(const f32x4& v, int i, int j) const {
    return (f32x4){v[0] * sam[i] * s4[j][0], v[1] * sam[i] * s4[j][1], v[2] * sam[i] * s4[j][2], v[3] * sam[i] * s4[j][3]};
  }
};

// 128 x 128 x 128 tile, 4 waves (2 x 2) x 64 x 64, two workgroups per CU (2 x 32 KiB LDS): the small-grid / remainder
// companion of the 256x256 kernel (ViT-sized problems, the ragged last columns of the gate/up projection).
__global__ __launch_bounds__(256, 2) void gemm_fp8_128x128_kernel(GemmF8Args p) {
  __shared__ __attribute__((aligned(16))) char lds[2 * 256 * F8_BK];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, h = lane >> 4;
  const int nwg = p.tiles_m * p.tiles_n;
  const int id = xcd_remap(blockIdx.x, nwg);
  const int tn = id / p.tiles_m, tm = id - tn * p.tiles_m;
  const int m0 = tm * 128, n0 = tn * 128;
  uint32_t a_off[4], w_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = i * 256 + tid;
    const int row = c >> 3;
    const int ch = (c & 7) ^ F8_SWZ(row);
    a_off[i] = (uint32_t)(min(m0 + row, p.M - 1) - m0) * (uint32_t)p.lda + ch * 16;
    w_off[i] = (uint32_t)(min(n0 + row, p.N - 1) - n0) * (uint32_t)p.ldw + ch * 16;
  }
  const char* a_base = (const char*)(p.A + (size_t)m0 * p.lda);
  const char* w_base = (const char*)(p.W + (size_t)n0 * p.ldw);
  const int wave_base = __builtin_amdgcn_readfirstlane(tid >> 6) * 1024;
  constexpr int A_BYTES = 128 * F8_BK;        // 16 KiB
  constexpr int BUF_BYTES = 256 * F8_BK;      // 32 KiB
  auto stage = [&](int buf) {
    char* base = lds + buf * BUF_BYTES + wave_base;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_base + a_off[i]),
                                       (__attribute__((address_space(3))) void*)(base + i * 4096), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_base + w_off[i]),
                                       (__attribute__((address_space(3))) void*)(base + A_BYTES + i * 4096), 16, 0, 0);
    a_base += F8_BK;
    w_base += F8_BK;
  };
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int sw = F8_SWZ(l15);
  const int rdlo = l15 * 128 + (((2 * h) ^ sw) << 4);
  const int rdhi = l15 * 128 + (((2 * h + 1) ^ sw) << 4);
  const int a_rd = wm * 64 * 128;
  const int w_rd = A_BYTES + wn * 64 * 128;
  auto frag = [&](const char* p0) -> i32x8 {
    const u32x4 lo = *(const u32x4*)(p0 + rdlo), hi = *(const u32x4*)(p0 + rdhi);
    return (i32x8){(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
  };
  const int nk = p.K / F8_BK;
  stage(0);
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) stage(cur ^ 1);
    const char* base = lds + cur * BUF_BYTES;
    i32x8 af[4], wf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) wf[j] = frag(base + w_rd + j * 2048);
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = frag(base + a_rd + i * 2048);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[j], af[i], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0,
                                                                     0x7f7f7f7f);
    __syncthreads();  // also drains the in-flight global_load_lds (vmcnt(0))
    cur ^= 1;
  }
  if (p.wide) {   // the loop's last __syncthreads: every read of the operand buffers is done, no LDS-DMA in flight
    gemm_epilogue_wide_dispatch<4, 4>(p, acc, m0 + wm * 64, n0 + wn * 64, lane, lds + wave * 16384,
                                      F8Scale<4>(p, m0 + wm * 64, n0 + wn * 64, l15, h));
    return;
  }
  f8_epilogue(p, acc, m0 + wm * 64, n0 + wn * 64, l15, h);
}

__global__ __launch_bounds__(512, 2) void gemm_fp8_256x256_kernel(GemmF8Args p) {
  extern __shared__ __attribute__((aligned(16))) char lds8[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int l15 = lane & 15, h = lane >> 4;

  const int nwg = p.tiles_m * p.tiles_n;
  const int id = xcd_remap(blockIdx.x, nwg);
  const int tn = id / p.tiles_m, tm = id - tn * p.tiles_m;
  const int m0 = tm * F8_B, n0 = tn * F8_B;

  uint32_t a_off[4], w_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = i * 512 + tid;
    const int row = c >> 3;
    const int ch = (c & 7) ^ F8_SWZ(row);
    a_off[i] = (uint32_t)(min(m0 + row, p.M - 1) - m0) * (uint32_t)p.lda + ch * 16;
    w_off[i] = (uint32_t)(min(n0 + row, p.N - 1) - n0) * (uint32_t)p.ldw + ch * 16;
  }
  const int nk_all = p.K / F8_BK;
  const int kt0 = p.part ? (int)((long long)nk_all * blockIdx.y / p.ksplit) : 0;
  const int kt1 = p.part ? (int)((long long)nk_all * (blockIdx.y + 1) / p.ksplit) : nk_all;
  const char* a_base = (const char*)(p.A + (size_t)m0 * p.lda) + (size_t)kt0 * F8_BK;
  const char* w_base = (const char*)(p.W + (size_t)n0 * p.ldw) + (size_t)kt0 * F8_BK;
  const int wave_base = __builtin_amdgcn_readfirstlane(tid >> 6) * 1024;
  constexpr int A_BYTES = F8_B * F8_BK;  // 32 KiB

  auto stage = [&](int buf, int kt) {
    char* base = lds8 + buf * F8_STAGE_BYTES + wave_base;
    const char* a_k = a_base + kt * F8_BK;
    const char* w_k = w_base + kt * F8_BK;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_k + a_off[i]),
                                       (__attribute__((address_space(3))) void*)(base + i * 8192), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_k + w_off[i]),
                                       (__attribute__((address_space(3))) void*)(base + A_BYTES + i * 8192), 16, 0,
                                       0);
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int sw = F8_SWZ(l15);
  const int rdlo = l15 * 128 + (((2 * h) ^ sw) << 4);
  const int rdhi = l15 * 128 + (((2 * h + 1) ^ sw) << 4);
  const int a_rd = wm * 128 * 128;           // + i * 2048
  const int w_rd = A_BYTES + wn * 64 * 128;  // + j * 2048

  auto frag = [&](const char* p0) -> i32x8 {
    const u32x4 lo = *(const u32x4*)(p0 + rdlo), hi = *(const u32x4*)(p0 + rdhi);
    return (i32x8){(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
  };

  const int nk = kt1 - kt0;
  stage(0, 0);
  stage(1, min(1, nk - 1));
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  int buf = 0;
  i32x8 af[4];   // A rows 0..63 of the wave's tile for the CURRENT K-step (read one half-step ahead)
#pragma unroll
  for (int i = 0; i < 4; ++i) af[i] = frag(lds8 + a_rd + i * 2048);
  for (int kt = 0; kt < nk; ++kt) {
    const char* base = lds8 + buf * F8_STAGE_BYTES;
    i32x8 wf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) wf[j] = frag(base + w_rd + j * 2048);
    __builtin_amdgcn_s_setprio(1);
    // phase A: column-major over the W fragments, so the first MFMAs start as soon as wf[0] has landed
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[j], af[i], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0,
                                                                     0x7f7f7f7f);
    i32x8 ag[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) ag[i] = frag(base + a_rd + (4 + i) * 2048);
    __builtin_amdgcn_s_setprio(0);
    // mid: every read of this buffer has landed in registers; the next stage has landed; buffer is dead
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 2 < nk) stage(buf, kt + 2);
    __builtin_amdgcn_s_setprio(1);
    // phase B: rows 64..127; the next K-step's rows 0..63 are read underneath (af is free since phase A)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        acc[4 + i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[j], ag[i], acc[4 + i][j], 0, 0, 0,
                                                                         0x7f7f7f7f, 0, 0x7f7f7f7f);
    if (kt + 1 < nk) {
      const char* nb = lds8 + (buf ^ 1) * F8_STAGE_BYTES;
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = frag(nb + a_rd + i * 2048);
    }
    __builtin_amdgcn_s_setprio(0);
    buf ^= 1;
  }

  if (p.part) {  // split-K: raw (unscaled) f32 partial sums; the finalize kernel applies scales and the epilogue
    float* dst = p.part + (size_t)blockIdx.y * p.M * p.N;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int m = m0 + wm * 128 + i * 16 + l15;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn * 64 + j * 16 + 4 * h;
        if (n < p.N) *(f32x4*)(dst + (size_t)m * p.N + n) = acc[i][j];
      }
    }
    return;
  }
  if (p.wide) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // slower waves may still be reading fragments of the last K-step
    gemm_epilogue_wide_dispatch<8, 4>(p, acc, m0 + wm * 128, n0 + wn * 64, lane, lds8 + wave * 16384,
                                      F8Scale<8>(p, m0 + wm * 128, n0 + wn * 64, l15, h));
    return;
  }
  // epilogue: two 64-row halves through the shared 4x4 epilogue
#pragma unroll
  for (int hm = 0; hm < 2; ++hm) {
    f32x4 sub[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) sub[i][j] = acc[hm * 4 + i][j];
    f8_epilogue(p, sub, m0 + wm * 128 + hm * 64, n0 + wn * 64, l15, h);
  }
}

// The ping-pong schedule of gemm_bf16_256x256_pp_kernel (gemm_bf16.hip: half-tile layout, stage / read / wait table and
// the hazard argument are written out there) on the fp8 MFMA: the tile row is again 128 bytes per K-step (128 e4m3
// elements), a fragment is two 16-byte chunks (2h, 2h+1) per lane, a phase is 8 MFMAs of K = 128 - the same LDS
// traffic, the same LDS-DMA count and the same MFMA time per phase as the bf16 kernel.
#define F8P_HALF_BYTES 16384
#define F8P_BUF_BYTES 65536
#define F8P_B0 0
#define F8P_A0 1
#define F8P_B1 2
#define F8P_A1 3

__global__ __launch_bounds__(512, 2) void gemm_fp8_256x256_pp_kernel(GemmF8Args p) {
  extern __shared__ __attribute__((aligned(16))) char lds9[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int l15 = lane & 15, h = lane >> 4;

  const int nwg = p.tiles_m * p.tiles_n;
  const int id = xcd_remap(blockIdx.x, nwg);
  const int tn = id / p.tiles_m, tm = id - tn * p.tiles_m;
  const int m0 = tm * F8_B, n0 = tn * F8_B;

  // The tile's 256 row scales and 256 column scales go to LDS behind the operand buffers by LDS-DMA (4 bytes per lane, the
  // oldest entry of the prologue's vmcnt queue): the epilogue then starts without a global round trip of its own
  // (tools/gemm_kscan.py KS_FP8=1: the fixed cost per round of tiles is what a 10-step K = 1280 tower projection feels).
  float* const lds_sc = (float*)(lds9 + 2 * F8P_BUF_BYTES);
  if (!p.part) {
    const float* src = (tid < 256) ? p.sa + min(m0 + tid, p.M - 1) : p.sw + min(n0 + tid - 256, p.N - 1);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(lds_sc + wave * 64), 4, 0, 0);
  }

  uint32_t a_off[2][2], w_off[2][2];   // [half][instruction]
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = i * 512 + tid;
    const int hr = c >> 3;
    const int ch = (c & 7) ^ F8_SWZ(hr);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int arow = (hr >> 6) * 128 + q * 64 + (hr & 63);
      const int wrow = (hr >> 5) * 64 + q * 32 + (hr & 31);
      a_off[q][i] = (uint32_t)(min(m0 + arow, p.M - 1) - m0) * (uint32_t)p.lda + ch * 16;
      w_off[q][i] = (uint32_t)(min(n0 + wrow, p.N - 1) - n0) * (uint32_t)p.ldw + ch * 16;
    }
  }
  const int nk_all = p.K / F8_BK;
  const int kt0 = p.part ? (int)((long long)nk_all * blockIdx.y / p.ksplit) : 0;
  const int kt1 = p.part ? (int)((long long)nk_all * (blockIdx.y + 1) / p.ksplit) : nk_all;
  const int nk = kt1 - kt0;
  const char* a_base = (const char*)(p.A + (size_t)m0 * p.lda) + (size_t)kt0 * F8_BK;
  const char* w_base = (const char*)(p.W + (size_t)n0 * p.ldw) + (size_t)kt0 * F8_BK;
  char* const wave_lds = lds9 + wave * 1024;

  auto stage = [&](int buf, int slot, int kt) {
    kt = min(kt, nk - 1);
    const bool is_a = slot & 1;
    const int q = slot >> 1;
    const char* src = (is_a ? a_base : w_base) + kt * F8_BK;
    char* dst = wave_lds + buf * F8P_BUF_BYTES + slot * F8P_HALF_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(src + (is_a ? a_off[q][i] : w_off[q][i])),
          (__attribute__((address_space(3))) void*)(dst + i * 8192), 16, 0, 0);
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int sw = F8_SWZ(l15);
  const int rd_lo = l15 * 128 + (((2 * h) ^ sw) << 4);
  const int rd_hi = l15 * 128 + (((2 * h + 1) ^ sw) << 4);
  const int a_rd = wr * (64 * 128);   // + ii * 2048
  const int b_rd = wc * (32 * 128);   // + jj * 2048
  u32x4 af[4][2], bq0[2][2], bq1[2][2];   // [fragment][lo / hi chunk]
  auto read_a = [&](int buf, int qa) {
    const char* base = lds9 + buf * F8P_BUF_BYTES + (qa ? F8P_A1 : F8P_A0) * F8P_HALF_BYTES + a_rd;
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
      af[ii][0] = *(const u32x4*)(base + ii * 2048 + rd_lo);
      af[ii][1] = *(const u32x4*)(base + ii * 2048 + rd_hi);
    }
  };
  auto read_b = [&](u32x4 (&bq)[2][2], int buf, int qb) {
    const char* base = lds9 + buf * F8P_BUF_BYTES + (qb ? F8P_B1 : F8P_B0) * F8P_HALF_BYTES + b_rd;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      bq[jj][0] = *(const u32x4*)(base + jj * 2048 + rd_lo);
      bq[jj][1] = *(const u32x4*)(base + jj * 2048 + rd_hi);
    }
  };
  auto cat = [](const u32x4& lo, const u32x4& hi) -> i32x8 {
    return (i32x8){(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
  };
#define F8P_BAR()                                 \
  do {                                            \
    __builtin_amdgcn_sched_barrier(0);            \
    asm volatile("s_barrier" ::: "memory");       \
    __builtin_amdgcn_sched_barrier(0);            \
  } while (0)
#define F8P_MFMA(QA, QB, BQ)                                                                                  \
  do {                                                                                                        \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                        \
    __builtin_amdgcn_s_setprio(1);                                                                            \
    _Pragma("unroll") for (int ii = 0; ii < 4; ++ii)                                                          \
    _Pragma("unroll") for (int jj = 0; jj < 2; ++jj)                                                          \
      acc[(QA) * 4 + ii][(QB) * 2 + jj] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(                   \
          cat(BQ[jj][0], BQ[jj][1]), cat(af[ii][0], af[ii][1]), acc[(QA) * 4 + ii][(QB) * 2 + jj], 0, 0, 0,   \
          0x7f7f7f7f, 0, 0x7f7f7f7f);                                                                         \
    __builtin_amdgcn_s_setprio(0);                                                                            \
  } while (0)
#define F8P_TILE(B, S1B, S1S, S1T, S2B, S2S, S2T, S3B, S3S, S3T, S4B, S4S, S4T)     \
  do {                                                                              \
    read_b(bq0, B, 0);                                                              \
    __builtin_amdgcn_sched_barrier(0);                                              \
    read_a(B, 0);                                                                   \
    stage(S1B, S1S, S1T);                                                           \
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");                              \
    F8P_BAR();                                                                      \
    F8P_MFMA(0, 0, bq0);                                                            \
    F8P_BAR();                                                                      \
    read_b(bq1, B, 1);                                                              \
    stage(S2B, S2S, S2T);                                                           \
    F8P_BAR();                                                                      \
    F8P_MFMA(0, 1, bq1);                                                            \
    F8P_BAR();                                                                      \
    read_a(B, 1);                                                                   \
    stage(S3B, S3S, S3T);                                                           \
    F8P_BAR();                                                                      \
    F8P_MFMA(1, 1, bq1);                                                            \
    F8P_BAR();                                                                      \
    stage(S4B, S4S, S4T);                                                           \
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                \
    F8P_BAR();                                                                      \
    F8P_MFMA(1, 0, bq0);                                                            \
    F8P_BAR();                                                                      \
  } while (0)

  stage(0, F8P_B0, 0); stage(0, F8P_A0, 0); stage(0, F8P_B1, 0); stage(0, F8P_A1, 0);
  stage(1, F8P_B0, 1); stage(1, F8P_A0, 1); stage(1, F8P_B1, 1);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  F8P_BAR();
  if (wr == 1) F8P_BAR();   // stagger: group 1 runs one barrier behind group 0

  int t = 0;
  for (; t + 1 < nk; t += 2) {
    F8P_TILE(0, 1, F8P_A1, t + 1, 0, F8P_B0, t + 2, 0, F8P_A0, t + 2, 0, F8P_B1, t + 2);
    F8P_TILE(1, 0, F8P_A1, t + 2, 1, F8P_B0, t + 3, 1, F8P_A0, t + 3, 1, F8P_B1, t + 3);
  }
  if (t < nk) F8P_TILE(0, 1, F8P_A1, t + 1, 0, F8P_B0, t + 2, 0, F8P_A0, t + 2, 0, F8P_B1, t + 2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // redundant trailing stages
  if (wr == 0) F8P_BAR();   // every wave has now executed the same number of barriers
#undef F8P_TILE
#undef F8P_MFMA
#undef F8P_BAR

  if (p.part) {
    float* dst = p.part + (size_t)blockIdx.y * p.M * p.N;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int m = m0 + wr * 128 + i * 16 + l15;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + wc * 64 + j * 16 + 4 * h;
        if (n < p.N) *(f32x4*)(dst + (size_t)m * p.N + n) = acc[i][j];
      }
    }
    return;
  }
  if (p.wide) {
    // as in gemm_bf16_256x256_pp_kernel: own LDS-DMA drained (vmcnt(0) above), one more barrier so that no wave stages
    // over a half-tile another wave's trailing stage is still landing in or a slower wave has yet to read
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const F8Scale<8> sc(lds_sc + wr * 128, lds_sc + 256 + wc * 64, l15, h);   // (behind the buffers: staging does not touch them)
    gemm_epilogue_wide_dispatch<8, 4>(p, acc, m0 + wr * 128, n0 + wc * 64, lane, lds9 + wave * 16384, sc);
    return;
  }
#pragma unroll
  for (int hm = 0; hm < 2; ++hm) {
    f32x4 sub[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) sub[i][j] = acc[hm * 4 + i][j];
    f8_epilogue(p, sub, m0 + wr * 128 + hm * 64, n0 + wc * 64, l15, h);
  }
}

// split-K second half: C = act((sum_slices part) * sa[m] * sw[n] + bias) + R
__global__ __launch_bounds__(256) void gemm_fp8_splitk_finalize_kernel(GemmF8Args p) {
  const int chunks = p.N >> 3;
  const long long total = (long long)p.M * chunks;
  const size_t slice = (size_t)p.M * p.N;
  for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < total; it += (long long)gridDim.x * 256) {
    const int m = (int)(it / chunks), n = (int)(it - (long long)m * chunks) * 8;
    const float* src = p.part + (size_t)m * p.N + n;
    float v[8];
    *(f32x4*)v = *(const f32x4*)src;
    *(f32x4*)(v + 4) = *(const f32x4*)(src + 4);
    for (int ks = 1; ks < p.ksplit; ++ks) {
      const f32x4 a = *(const f32x4*)(src + ks * slice), b = *(const f32x4*)(src + ks * slice + 4);
      v[0] += a[0]; v[1] += a[1]; v[2] += a[2]; v[3] += a[3];
      v[4] += b[0]; v[5] += b[1]; v[6] += b[2]; v[7] += b[3];
    }
    const float sam = p.sa[m];
    const f32x4 s0 = *(const f32x4*)(p.sw + n), s1 = *(const f32x4*)(p.sw + n + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] *= sam * s0[e]; v[4 + e] *= sam * s1[e]; }
    if (p.bias) {
      float f[8];
      unpack8(*(const u32x4*)(p.bias + n), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += f[e];
    }
    if (p.act != F8_ACT_NONE) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = f8_act(v[e], p.act);
    }
    if (p.R) {
      float f[8];
      unpack8(*(const u32x4*)(p.R + (size_t)m * p.ldr + n), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += f[e];
    }
    *(u32x4*)(p.C + (size_t)m * p.ldc + n) = pack8(v);
  }
}

// work != NULL with ksplit in 2..8: split-K (f32 partials in `work`, ksplit*M*N floats; needs N % 8 == 0, no SwiGLU)
extern "C" int vis_gemm_fp8(const void* Aq, const void* sa, const void* Wq, const void* sw, const void* bias,
                            const void* R, void* C, void* work, int ksplit, int M, int N, int K, int lda, int ldw,
                            int ldc, int ldr, int act, hipStream_t stream) {
  if (!Aq || !sa || !Wq || !sw || !C || M <= 0 || N <= 0 || K <= 0) return VIS_ERR_ARG;
  if (K % F8_BK != 0 || N % 4 != 0 || lda % 16 != 0 || ldw % 16 != 0 || lda < K || ldw < K) return VIS_ERR_ARG;
  if (ldc % 4 != 0 || (R && ldr % 4 != 0)) return VIS_ERR_ARG;
  if (act < F8_ACT_NONE || act > F8_ACT_SWIGLU) return VIS_ERR_ARG;
  if (act == F8_ACT_SWIGLU && (N % 32 != 0 || bias || R)) return VIS_ERR_ARG;
  if (((uintptr_t)Aq | (uintptr_t)Wq | (uintptr_t)sw) & 15) return VIS_ERR_ARG;
  if (((uintptr_t)C | (uintptr_t)bias | (uintptr_t)R) & 7 || ((uintptr_t)sa & 3)) return VIS_ERR_ARG;
  static const bool attr_ok = [] {
    return hipFuncSetAttribute((const void*)gemm_fp8_256x256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                               F8_LDS_BYTES) == hipSuccess &&
           hipFuncSetAttribute((const void*)gemm_fp8_256x256_pp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                               F8_LDS_BYTES) == hipSuccess;
  }();
  static const int pp_env = [] { const char* e = getenv("VIS_GEMM_PP"); return e ? atoi(e) : 1; }();   // 0: 2-phase kernel (A/B)
  auto* const k256 = pp_env ? gemm_fp8_256x256_pp_kernel : gemm_fp8_256x256_kernel;
  if (!attr_ok) return VIS_ERR_LAUNCH;
  GemmF8Args p;
  p.A = (const uint8_t*)Aq; p.W = (const uint8_t*)Wq; p.sa = (const float*)sa; p.sw = (const float*)sw;
  p.bias = (const bf16_t*)bias; p.R = (const bf16_t*)R; p.C = (bf16_t*)C;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.ldr = ldr; p.act = act;
  p.tiles_m = (M + F8_B - 1) / F8_B;
  p.tiles_n = (N + F8_B - 1) / F8_B;
  p.part = nullptr; p.ksplit = 1;
  static const int wide_env = [] { const char* e = getenv("VIS_GEMM_WIDE"); return e ? atoi(e) : 1; }();   // 0: direct epilogue (A/B)
  p.wide = wide_env && N % 8 == 0 && ldc % 8 == 0 && (!R || ldr % 8 == 0) && !(((uintptr_t)C | (uintptr_t)R) & 15) &&
           !(act == F8_ACT_SWIGLU && N % 16 != 0);
  static const int nt_env = [] { const char* e = getenv("VIS_GEMM_NT"); return e ? atoi(e) : 1; }();
  p.nt = nt_env == 2 || (nt_env == 1 && (size_t)M * (act == F8_ACT_SWIGLU ? N / 2 : N) * 2 >= ((size_t)64 << 20));
  if (work) {
    if (ksplit < 2 || ksplit > 8 || K / F8_BK < 2 * ksplit || N % 8 != 0 || ldc % 8 != 0 || (R && ldr % 8 != 0) ||
        act == F8_ACT_SWIGLU || ((uintptr_t)work & 15) || (((uintptr_t)C | (uintptr_t)bias | (uintptr_t)R) & 15))
      return VIS_ERR_ARG;
    p.part = (float*)work; p.ksplit = ksplit;
  }
  vis_clear_error();
  // tile choice by a cost model in us, fitted to tools/gemm_kscan.py KS_FP8=1 on the r05 kernels (VIS_GEMM8_TILE=1|4 forces):
  //   256 x 256 ping-pong, one workgroup per CU : a round of 256 tiles costs 8.2 + 1.27 per 128-wide K-step
  //   128 x 128, two workgroups per CU          : a round of 512 tiles costs 4.5 + 1.06 per K-step
  // candidates: everything on one kernel, or whole rounds of the big tile + the remaining columns on the small one (a
  // second launch: + 2 us).  r04's rule ("256 only when the last round is at least half full") sent the LLM's qkv / o
  // projections at 4 images (378 / 294 big tiles) to the small kernel: 121 / 118 us where two big rounds take ~88.
  static const int forced = [] { const char* e = getenv("VIS_GEMM8_TILE"); return e ? atoi(e) : 0; }();
  const int t4 = p.tiles_m * p.tiles_n;
  const int nk = K / F8_BK;
  const float c256 = 8.2f + 1.27f * nk, c128 = 4.5f + 1.06f * nk;
  auto rounds = [](long long tiles, int per) { return (float)((tiles + per - 1) / per); };
  auto tiles128 = [&](int n) { return (long long)((M + 127) / 128) * ((n + 127) / 128); };
  const float cost_big = rounds(t4, 256) * c256, cost_small = rounds(tiles128(N), 512) * c128;
  const int cols4 = (t4 / 256) * 256 / p.tiles_m;  // columns of big tiles in whole rounds only
  const int n_off = cols4 * F8_B;
  const float cost_mixed = (cols4 > 0 && n_off < N)
                               ? rounds((long long)cols4 * p.tiles_m, 256) * c256 + rounds(tiles128(N - n_off), 512) * c128 + 2.f
                               : 1e30f;
  int choice;   // 4: big, 1: small, 5: mixed
  if (work || forced == 4) choice = 4;
  else if (forced == 1 || M < 1024) choice = 1;
  else choice = (cost_big <= cost_small && cost_big <= cost_mixed) ? 4 : (cost_mixed < cost_small ? 5 : 1);
  auto launch128 = [&](GemmF8Args q) {
    q.tiles_m = (q.M + 127) / 128;
    q.tiles_n = (q.N + 127) / 128;
    hipLaunchKernelGGL(gemm_fp8_128x128_kernel, dim3(q.tiles_m * q.tiles_n), dim3(256), 0, stream, q);
  };
  if (choice == 4) {
    hipLaunchKernelGGL(k256, dim3(t4, p.ksplit), dim3(512), F8_LDS_BYTES, stream, p);
  } else if (choice == 5) {
    GemmF8Args q = p;
    q.N = cols4 * F8_B;
    q.tiles_n = cols4;
    hipLaunchKernelGGL(k256, dim3(q.tiles_m * q.tiles_n, 1), dim3(512), F8_LDS_BYTES, stream, q);
    const int c_off = (act == F8_ACT_SWIGLU) ? n_off / 2 : n_off;
    GemmF8Args r = p;
    r.W += (size_t)n_off * ldw;
    r.sw += n_off;
    if (r.bias) r.bias += n_off;
    if (r.R) r.R += c_off;
    r.C += c_off;
    r.N = N - n_off;
    launch128(r);
  } else {
    launch128(p);
  }
  if (work) {
    const long long total = (long long)M * (N / 8);
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(gemm_fp8_splitk_finalize_kernel, dim3(blocks), dim3(256), 0, stream, p);
  }
  return vis_check_launch();
}

// ---------------------------------------------------------------------------
// vis_quant_rows_fp8: per-row (per-token) dynamic quantisation of bf16 activations to e4m3:
//   scale[m] = max(|x[m][:]|) / 448 (>= 1e-12),  q[m][k] = e4m3_rne(x[m][k] / scale[m])
// with an optional fused norm in front, producing exactly the bf16 values vis_rmsnorm_bf16 / vis_layernorm_bf16 would
// have written: norm_w only -> RMSNorm (x <- bf16(bf16(x * rstd) * w)); norm_w and norm_b -> LayerNorm.  One wave per row, the row lives in registers.
#define QR_MAX_CHUNKS 8  // register-resident rows: K <= 64 * 8 * 8 = 4096 (the fused-norm case); longer rows stream

__device__ __forceinline__ u32x2 qr_pack8(const float* f, float inv) {
  // v_cvt_pk_fp8_f32: two floats -> two e4m3 bytes (RNE, saturating), into the low or high half of a dword
  int w0 = 0, w1 = 0;
  w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0] * inv, f[1] * inv, w0, false);
  w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2] * inv, f[3] * inv, w0, true);
  w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4] * inv, f[5] * inv, w1, false);
  w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6] * inv, f[7] * inv, w1, true);
  return (u32x2){(uint32_t)w0, (uint32_t)w1};
}

template <int QCH>   // 16-byte chunks per lane the register path holds: 3 (rows up to 1536 elements: the ViT's LayerNorm rows at a third of
                     // the registers, r05) or QR_MAX_CHUNKS
__global__ __launch_bounds__(256) void quant_rows_fp8_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ nw,
                                                             const bf16_t* __restrict__ nb, uint8_t* __restrict__ q,
                                                             float* __restrict__ scale, int rows, int K, int ldx,
                                                             int ldq, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nch = K >> 3;
  const bf16_t* xr = x + (size_t)row * ldx;
  uint8_t* qr = q + (size_t)row * ldq;
  if (nch <= 64 * QCH) {  // the row lives in registers (uniform branch)
    float v[QCH][8];
    float ss = 0.f;
    {   // r05: every load unconditional (clamped chunk) and issued before the first use - a guarded load drains the queue
      u32x4 raw[QCH];
#pragma unroll
      for (int i = 0; i < QCH; ++i) raw[i] = *(const u32x4*)(xr + min(lane + i * 64, nch - 1) * 8);
#pragma unroll
      for (int i = 0; i < QCH; ++i) {
        if (lane + i * 64 < nch) {
          unpack8(raw[i], v[i]);
#pragma unroll
          for (int e = 0; e < 8; ++e) ss += v[i][e] * v[i][e];
        }
      }
    }
    if (nw && nb) {   // LayerNorm (ViT): same arithmetic and single bf16 rounding as norm_rows_kernel<true>
      float sm = 0.f;
#pragma unroll
      for (int i = 0; i < QCH; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
#pragma unroll
          for (int e = 0; e < 8; ++e) sm += v[i][e];
        }
      }
      const float mean = wave_sum(sm) / (float)K;
      float d2 = 0.f;
#pragma unroll
      for (int i = 0; i < QCH; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
#pragma unroll
          for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; d2 += d * d; }
        }
      }
      const float rstd = rsqrtf(wave_sum(d2) / (float)K + eps);
#pragma unroll
      for (int i = 0; i < QCH; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
          float w[8], b[8];
          unpack8(*(const u32x4*)(nw + c * 8), w);
          unpack8(*(const u32x4*)(nb + c * 8), b);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[i][e] = bf2f(f2bf((v[i][e] - mean) * rstd * w[e] + b[e]));
        }
      }
    } else if (nw) {  // RMSNorm (LLM)
      ss = wave_sum(ss);
      const float rstd = rsqrtf(ss / (float)K + eps);
#pragma unroll
      for (int i = 0; i < QCH; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
          float w[8];
          unpack8(*(const u32x4*)(nw + c * 8), w);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[i][e] = bf2f(f2bf(bf2f(f2bf(v[i][e] * rstd)) * w[e]));
        }
      }
    }
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < QCH; ++i) {
      const int c = lane + i * 64;
      if (c < nch) {
#pragma unroll
        for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(v[i][e]));
      }
    }
    amax = wave_max(amax);
    const float sc = fmaxf(amax / 448.0f, 1e-12f);
    const float inv = 1.0f / sc;
    if (lane == 0) scale[row] = sc;
#pragma unroll
    for (int i = 0; i < QCH; ++i) {
      const int c = lane + i * 64;
      if (c < nch) *(u32x2*)(qr + c * 8) = qr_pack8(v[i], inv);
    }
  } else {  // long rows (no norm): max pass, then a second pass over the (L2-resident) row
    float amax = 0.f;
    for (int c = lane; c < nch; c += 64) {
      float f[8];
      unpack8(*(const u32x4*)(xr + c * 8), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(f[e]));
    }
    amax = wave_max(amax);
    const float sc = fmaxf(amax / 448.0f, 1e-12f);
    const float inv = 1.0f / sc;
    if (lane == 0) scale[row] = sc;
    for (int c = lane; c < nch; c += 64) {
      float f[8];
      unpack8(*(const u32x4*)(xr + c * 8), f);
      *(u32x2*)(qr + c * 8) = qr_pack8(f, inv);
    }
  }
}

// Rows without a fused norm, up to 64 * 8 * NCH elements (r05): the row stays in registers as packed bf16 - NCH 16-byte loads per
// lane, all issued before the first use - so it is read ONCE; the streaming path of quant_rows_fp8_kernel read it twice through a
// load-per-iteration loop and ran the activation rows of the fp8 prompt pass (ViT fc1 output 19600 x 5120, SwiGLU output
// 5156 x 18944: ~300 MB each) at 2.7 TB/s.  Same arithmetic and bytes as that path.
template <int NCH>
__global__ __launch_bounds__(256) void quant_rows_fp8_wide_kernel(const bf16_t* __restrict__ x, uint8_t* __restrict__ q,
                                                                  float* __restrict__ scale, int rows, int K, int ldx, int ldq) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nch = K >> 3;
  const bf16_t* xr = x + (size_t)row * ldx;
  uint8_t* qr = q + (size_t)row * ldq;
  u32x4 raw[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) raw[i] = *(const u32x4*)(xr + min(lane + i * 64, nch - 1) * 8);   // (a clamped duplicate changes no maximum)
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    float f[8];
    unpack8(raw[i], f);
#pragma unroll
    for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(f[e]));
  }
  amax = wave_max(amax);
  const float sc = fmaxf(amax / 448.0f, 1e-12f);
  const float inv = 1.0f / sc;
  if (lane == 0) scale[row] = sc;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + i * 64;
    if (c < nch) {
      float f[8];
      unpack8(raw[i], f);
      *(u32x2*)(qr + c * 8) = qr_pack8(f, inv);
    }
  }
}

// The same for very long rows (SwiGLU output, 18944 elements): one WORKGROUP per row, NCHT 16-byte loads per thread, the four waves'
// maxima meet in LDS.  A wave-per-row kernel needs 37 chunks per lane there = 256 VGPRs = two waves per SIMD, each loading its whole
// row before storing any of it: 94.9 us for 293 MB (3.1 TB/s) at the fp8 prompt pass's 5156 x 18944.
template <int NCHT>
__global__ __launch_bounds__(256) void quant_rows_fp8_rowwg_kernel(const bf16_t* __restrict__ x, uint8_t* __restrict__ q,
                                                                   float* __restrict__ scale, int K, int ldx, int ldq) {
  __shared__ float wmax[4];
  const int tid = threadIdx.x, row = blockIdx.x;
  const int nch = K >> 3;
  const bf16_t* xr = x + (size_t)row * ldx;
  uint8_t* qr = q + (size_t)row * ldq;
  u32x4 raw[NCHT];
#pragma unroll
  for (int i = 0; i < NCHT; ++i) raw[i] = *(const u32x4*)(xr + min(tid + i * 256, nch - 1) * 8);
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < NCHT; ++i) {
    float f[8];
    unpack8(raw[i], f);
#pragma unroll
    for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(f[e]));
  }
  amax = wave_max(amax);
  if ((tid & 63) == 0) wmax[tid >> 6] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
  const float sc = fmaxf(amax / 448.0f, 1e-12f);
  const float inv = 1.0f / sc;
  if (tid == 0) scale[row] = sc;
#pragma unroll
  for (int i = 0; i < NCHT; ++i) {
    const int c = tid + i * 256;
    if (c < nch) {
      float f[8];
      unpack8(raw[i], f);
      *(u32x2*)(qr + c * 8) = qr_pack8(f, inv);
    }
  }
}

extern "C" int vis_quant_rows_fp8(const void* x, const void* norm_w, const void* norm_b, void* q, void* scale,
                                  int rows, int K, int ldx, int ldq, float eps, hipStream_t stream) {
  if (!x || !q || !scale || rows <= 0 || K <= 0) return VIS_ERR_ARG;
  if (K % 8 != 0 || ldx % 8 != 0 || ldq % 8 != 0 || ldq < K) return VIS_ERR_ARG;
  if (norm_w && K > 64 * 8 * QR_MAX_CHUNKS) return VIS_ERR_ARG;  // the fused norm needs the row in registers
  if (norm_b && !norm_w) return VIS_ERR_ARG;
  if (((uintptr_t)x | (uintptr_t)norm_w | (uintptr_t)norm_b) & 15 || ((uintptr_t)q & 7) || ((uintptr_t)scale & 3))
    return VIS_ERR_ARG;
  vis_clear_error();
  static const int wide_env = [] { const char* e = getenv("VIS_QUANT_WIDE"); return e ? atoi(e) : 1; }();   // 0: r04 kernel (A/B)
  const int nch_lane = (K / 8 + 63) / 64;   // 16-byte chunks per lane
  const dim3 grid((rows + 3) / 4), block(256);
  if (!norm_w && wide_env && nch_lane > 4 && nch_lane <= 40) {
#define QR_WIDE(N) hipLaunchKernelGGL(quant_rows_fp8_wide_kernel<N>, grid, block, 0, stream, (const bf16_t*)x, (uint8_t*)q, (float*)scale, rows, K, ldx, ldq)
    if (nch_lane <= 8) QR_WIDE(8);
    else if (nch_lane <= 12) QR_WIDE(12);
    else if (wide_env == 2) { if (nch_lane <= 24) QR_WIDE(24); else QR_WIDE(40); }      // (A/B: a wave per row at any length)
    else {                                                                               // a workgroup per row
      const int ncht = (K / 8 + 255) / 256;                                              // 16-byte chunks per thread: 4 .. 10
#define QR_ROWWG(N) hipLaunchKernelGGL(quant_rows_fp8_rowwg_kernel<N>, dim3(rows), block, 0, stream, (const bf16_t*)x, (uint8_t*)q, (float*)scale, K, ldx, ldq)
      if (ncht <= 6) QR_ROWWG(6); else QR_ROWWG(10);
#undef QR_ROWWG
    }
#undef QR_WIDE
    return vis_check_launch();
  }
  if (wide_env && nch_lane <= 3)
    hipLaunchKernelGGL(quant_rows_fp8_kernel<3>, grid, block, 0, stream, (const bf16_t*)x,
                       (const bf16_t*)norm_w, (const bf16_t*)norm_b, (uint8_t*)q, (float*)scale, rows, K, ldx, ldq, eps);
  else
    hipLaunchKernelGGL(quant_rows_fp8_kernel<QR_MAX_CHUNKS>, grid, block, 0, stream, (const bf16_t*)x,
                       (const bf16_t*)norm_w, (const bf16_t*)norm_b, (uint8_t*)q, (float*)scale, rows, K, ldx, ldq, eps);
  return vis_check_launch();
}
