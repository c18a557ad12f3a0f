// K2: bf16 GEMM family for the Qwen2-VL prefill path on gfx950 (MI355X).
//
//   C[M,N] = act(A[M,K] * W[N,K]^T + bias[N]) + R[M,N]
//
// A is the row-major activation matrix, W is an nn.Linear weight ([out,in],
// K-contiguous), so both MFMA operands are K-contiguous rows ("NT" GEMM) and
// both are fetched with ds_read_b128.  Replaces the torch ops behind
// TF:models/qwen2_vl/modeling_qwen2_vl.py:251-274 (patch-embed conv == GEMM),
// :281-286 (merger), :296-298 (ViT MLP), :349-350 (ViT qkv/proj),
// :459-466 (LLM MLP incl. SwiGLU), :501-504 (LLM q/k/v/o).
//
// Design (CDNA4):
//  * 128x128x64 block tile, 256 threads = 4 waves (2x2), each wave a 64x64
//    output tile = 4x4 MFMA 16x16x32 bf16 tiles with f32 accumulators.
//  * global -> LDS with global_load_lds_dwordx4 (16 B/lane, no VGPR staging).
//    The LDS image is lane-linear, so the bank-conflict swizzle is applied on
//    the per-lane SOURCE address: physical 16-B chunk c of row r holds logical
//    chunk c ^ (r & 7); the ds_read applies the same XOR.  That makes every
//    ds_read_b128 16-lane group hit 16 distinct 16-B slots (conflict-free).
//  * double-buffered LDS (2 x 32 KiB) -> 2 workgroups per CU.
//  * the MFMA is issued with W as the A operand and the activations as the B
//    operand, so the accumulator holds 4 consecutive output COLUMNS per lane
//    (D[row=n][col=m]) and the epilogue stores 8 contiguous bytes per lane.
//  * fused epilogues: bias, QuickGELU, erf-GELU, residual add, SwiGLU over a
//    gate/up weight interleaved in 16-row groups.
//  * 1-D grid with a bijective XCD remap; all M-tiles that share one W panel
//    are adjacent inside one XCD so the panel is read from HBM once.
#include "common.hip.h"
#include <stdlib.h>

#define GEMM_BM 128
#define GEMM_BN 128
#define GEMM_BK 64

#include "gemm_epilogue.hip.h"


struct GemmArgs {
  const bf16_t* A;
  const bf16_t* W;
  const bf16_t* bias;  // [N] or null
  const bf16_t* R;     // [M, ldr] residual or null
  bf16_t* C;
  int M, N, K;
  int lda, ldw, ldc, ldr;
  int act;
  int tiles_m, tiles_n;
  // 32-row (batched decode) tile only: K split over grid.y, f32 partials part[ks][16][N]; out_f32: direct f32 C
  float* part;
  int ksplit, out_f32;
  int wide;   // host-checked alignment preconditions of gemm_epilogue_wide hold
  int nt;     // wide epilogue: non-temporal C stores (the output does not displace the operand tiles other workgroups re-read from L2)
};

// Epilogue shared by both tile shapes.  The MFMAs were issued with W as the A operand, so a lane holds
// D[n = nbase + 16 j + 4 h + r][m = mbase + 16 i + l15], r = 0..3: four consecutive output columns.
__device__ __forceinline__ void gemm_epilogue_n(const GemmArgs& p, f32x4 (&acc)[4][4], int mbase, int nbase, int l15,
                                                int h, int ntiles);

__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, f32x4 (&acc)[4][4], int mbase, int nbase, int l15,
                                              int h) {
  gemm_epilogue_n(p, acc, mbase, nbase, l15, h, 4);
}

// ntiles (<= 4): number of valid 16-column tiles in acc (compile-time constant at every call site)
__device__ __forceinline__ void gemm_epilogue_n(const GemmArgs& p, f32x4 (&acc)[4][4], int mbase, int nbase, int l15,
                                                int h, int ntiles) {
  const bool swiglu = (p.act == ACT_SWIGLU);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = mbase + i * 16 + l15;
    if (m >= p.M) continue;
    if (swiglu) {
#pragma unroll
      for (int j = 0; j < 4; j += 2) {
        const int n = nbase + j * 16 + 4 * h;  // gate row index in the interleaved weight
        if (j >= ntiles || n >= p.N) continue;
        const int oc = (nbase >> 1) + (j >> 1) * 16 + 4 * h;
        float v[4], bg[4] = {0.f, 0.f, 0.f, 0.f}, bu[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) {   // interleaved like the weight rows: gate biases at n .. n+3, up biases 16 further
          const u32x2 b0 = *(const u32x2*)(p.bias + n), b1 = *(const u32x2*)(p.bias + n + 16);
          bg[0] = __uint_as_float(b0[0] << 16); bg[1] = __uint_as_float(b0[0] & 0xffff0000u);
          bg[2] = __uint_as_float(b0[1] << 16); bg[3] = __uint_as_float(b0[1] & 0xffff0000u);
          bu[0] = __uint_as_float(b1[0] << 16); bu[1] = __uint_as_float(b1[0] & 0xffff0000u);
          bu[2] = __uint_as_float(b1[1] << 16); bu[3] = __uint_as_float(b1[1] & 0xffff0000u);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float g = acc[i][j][r] + bg[r], u = acc[i][j + 1][r] + bu[r];
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
        if (j >= ntiles || n >= p.N) continue;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r];
        if (p.bias) {
          const u32x2 b = *(const u32x2*)(p.bias + n);
          v[0] += __uint_as_float(b[0] << 16);
          v[1] += __uint_as_float(b[0] & 0xffff0000u);
          v[2] += __uint_as_float(b[1] << 16);
          v[3] += __uint_as_float(b[1] & 0xffff0000u);
        }
        if (p.act != ACT_NONE) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = act_apply(v[r], p.act);
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


__global__ __launch_bounds__(256, 2) void gemm_bf16_128x128_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) char lds[2 * (GEMM_BM + GEMM_BN) * GEMM_BK * 2];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, h = lane >> 4;

  const int nwg = p.tiles_m * p.tiles_n;
  const int id = xcd_remap(blockIdx.x, nwg);
  const int tn = id / p.tiles_m, tm = id - tn * p.tiles_m;
  const int m0 = tm * GEMM_BM, n0 = tn * GEMM_BN;

  // ---- staging addresses: 4 A chunks + 4 W chunks of 16 B per thread per K-step.
  // Keep the K-loop free of vector address arithmetic: a wave-uniform (SGPR) base pointer that advances by
  // one K-step with scalar adds, plus a per-lane 32-bit byte offset that never changes.
  uint32_t a_off[4], w_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = i * 256 + tid;
    const int row = c >> 3;
    const int ch = (c & 7) ^ (row & 7);  // logical chunk stored at this physical slot
    a_off[i] = (uint32_t)(min(m0 + row, p.M - 1) - m0) * (uint32_t)(p.lda * 2) + ch * 16;
    w_off[i] = (uint32_t)(min(n0 + row, p.N - 1) - n0) * (uint32_t)(p.ldw * 2) + ch * 16;
  }
  const char* a_base = (const char*)(p.A + (size_t)m0 * p.lda);
  const char* w_base = (const char*)(p.W + (size_t)n0 * p.ldw);
  const int wave_base = __builtin_amdgcn_readfirstlane(tid >> 6) * 1024;  // provably wave-uniform LDS slot
  constexpr int A_BYTES = GEMM_BM * GEMM_BK * 2;  // 16 KiB
  constexpr int BUF_BYTES = (GEMM_BM + GEMM_BN) * GEMM_BK * 2;

  auto stage = [&](int buf) {
    char* base = lds + buf * BUF_BYTES + wave_base;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_base + a_off[i]),
                                       (__attribute__((address_space(3))) void*)(base + i * 4096), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_base + w_off[i]),
                                       (__attribute__((address_space(3))) void*)(base + A_BYTES + i * 4096), 16, 0,
                                       0);
    a_base += GEMM_BK * 2;
    w_base += GEMM_BK * 2;
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // per-lane LDS read offsets (bytes) for k-substep 0/1; row&7 == lane&7
  const int sw = lane & 7;
  const int rd0 = l15 * 128 + (((0 + h) ^ sw) << 4);
  const int rd1 = l15 * 128 + (((4 + h) ^ sw) << 4);
  const int a_rd = wm * 64 * 128;            // + i*16*128
  const int w_rd = A_BYTES + wn * 64 * 128;  // + j*16*128

  const int nk = p.K / GEMM_BK;
  stage(0);
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) stage(cur ^ 1);
    const char* base = lds + cur * BUF_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int rd = ks ? rd1 : rd0;
      bf16x8 af[4], wf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = *(const bf16x8*)(base + a_rd + i * 2048 + rd);
#pragma unroll
      for (int j = 0; j < 4; ++j) wf[j] = *(const bf16x8*)(base + w_rd + j * 2048 + rd);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
    }
    __syncthreads();  // also drains the in-flight global_load_lds (vmcnt(0))
    cur ^= 1;
  }

  // the loop's last __syncthreads() retired every LDS read and every LDS-DMA: the buffers are free for staging
  if (p.wide) gemm_epilogue_wide_dispatch<4, 4>(p, acc, m0 + wm * 64, n0 + wn * 64, lane, lds + wave * 16384);
  else gemm_epilogue(p, acc, m0 + wm * 64, n0 + wn * 64, l15, h);
}

// ---------------------------------------------------------------------------
// 256x128x64 tile, 512 threads = 8 waves (4 x 2, each 64x64), ONE workgroup per CU, 3-stage LDS ring
// (3 x 48 KiB).  The global_load_lds of K-step t+2 are issued while step t is computed; the only waits in the
// loop are a COUNTED s_waitcnt vmcnt(6) (this thread's 6 loads of step t+1 may stay in flight) and one raw
// s_barrier per K-step - never a __syncthreads(), whose fence would drain the in-flight LDS-DMA.
//   iteration t:  wait(stage t landed) ; barrier ; issue stage t+2 into the buffer everyone just finished
//                 reading (stage t-1) ; ds_read + 32 MFMA on stage t
// The barrier both publishes stage t (every wave waited for its own part) and retires the reads of stage t-1.
#define GEMM2_BM 256
#define GEMM2_STAGE_BYTES ((GEMM2_BM + GEMM_BN) * GEMM_BK * 2)  // 49152
#define GEMM2_LDS_BYTES (3 * GEMM2_STAGE_BYTES)                  // 147456

__global__ __launch_bounds__(512, 2) void gemm_bf16_256x128_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char lds2[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, h = lane >> 4;

  const int nwg = p.tiles_m * p.tiles_n;
  const int id = xcd_remap(blockIdx.x, nwg);
  const int tn = id / p.tiles_m, tm = id - tn * p.tiles_m;
  const int m0 = tm * GEMM2_BM, n0 = tn * GEMM_BN;

  uint32_t a_off[4], w_off[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = i * 512 + tid;
    const int row = c >> 3;
    const int ch = (c & 7) ^ (row & 7);
    a_off[i] = (uint32_t)(min(m0 + row, p.M - 1) - m0) * (uint32_t)(p.lda * 2) + ch * 16;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = i * 512 + tid;
    const int row = c >> 3;
    const int ch = (c & 7) ^ (row & 7);
    w_off[i] = (uint32_t)(min(n0 + row, p.N - 1) - n0) * (uint32_t)(p.ldw * 2) + ch * 16;
  }
  const char* a_base = (const char*)(p.A + (size_t)m0 * p.lda);
  const char* w_base = (const char*)(p.W + (size_t)n0 * p.ldw);
  const int wave_base = __builtin_amdgcn_readfirstlane(tid >> 6) * 1024;
  constexpr int A_BYTES = GEMM2_BM * GEMM_BK * 2;  // 32 KiB

  auto stage = [&](int buf) {
    char* base = lds2 + buf * GEMM2_STAGE_BYTES + wave_base;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_base + a_off[i]),
                                       (__attribute__((address_space(3))) void*)(base + i * 8192), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_base + w_off[i]),
                                       (__attribute__((address_space(3))) void*)(base + A_BYTES + i * 8192), 16, 0,
                                       0);
    a_base += GEMM_BK * 2;
    w_base += GEMM_BK * 2;
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int sw = lane & 7;
  const int rd0 = l15 * 128 + (((0 + h) ^ sw) << 4);
  const int rd1 = l15 * 128 + (((4 + h) ^ sw) << 4);
  const int a_rd = wm * 64 * 128;
  const int w_rd = A_BYTES + wn * 64 * 128;

  // Software pipeline (fragments double-buffered in registers, one barrier per K-step placed BETWEEN the two
  // MFMA clusters so every LDS fragment read overlaps a cluster instead of stalling behind the barrier):
  //   F0 = frags(t, k-half 0) already in registers
  //   A:  ds_read F1 = frags(t, 1)          || 16 MFMA on F0
  //       s_waitcnt vmcnt(0) [stage t+1, issued a whole K-step ago] ; s_barrier ; issue stage t+2 -> buf (t+2)%3
  //   B:  ds_read F0 = frags(t+1, 0)        || 16 MFMA on F1
  // Buffer (t+2)%3 == (t-1)%3 was last read in iteration t-1 (its MFMAs have retired before this barrier).
  const int nk = p.K / GEMM_BK;
  stage(0);
  if (nk > 1) stage(1);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6) : "memory");
  if (nk == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  bf16x8 af0[4], wf0[4], af1[4], wf1[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) af0[i] = *(const bf16x8*)(lds2 + a_rd + i * 2048 + rd0);
#pragma unroll
  for (int j = 0; j < 4; ++j) wf0[j] = *(const bf16x8*)(lds2 + w_rd + j * 2048 + rd0);
  auto mfma16 = [&](bf16x8 (&wf)[4], bf16x8 (&af)[4]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  int buf = 0;
  // main loop: every K-step except the last; the body is branch-free apart from the (uniform) stage issue, so
  // the compiler's s_waitcnt lgkmcnt counts stay exact (fragment reads always overlap an MFMA cluster)
  for (int kt = 0; kt + 1 < nk; ++kt) {
    const char* base = lds2 + buf * GEMM2_STAGE_BYTES;
    const int nbuf = (buf == 2) ? 0 : buf + 1;
    // phase A.  F0 crossed the loop back-edge, so hipcc waits lgkmcnt(0) before its first use: issue the F1
    // reads only AFTER the first MFMA (order pinned), then they overlap the other 15 MFMAs of the cluster.
    __builtin_amdgcn_s_setprio(1);
    acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0[0], af0[0], acc[0][0], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) af1[i] = *(const bf16x8*)(base + a_rd + i * 2048 + rd1);
#pragma unroll
    for (int j = 0; j < 4; ++j) wf1[j] = *(const bf16x8*)(base + w_rd + j * 2048 + rd1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (i | j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0[j], af0[i], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // stage kt+1 (issued a whole K-step ago) has landed
    __builtin_amdgcn_s_barrier();
    if (kt + 2 < nk) stage((nbuf == 2) ? 0 : nbuf + 1);
    const char* nbase = lds2 + nbuf * GEMM2_STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) af0[i] = *(const bf16x8*)(nbase + a_rd + i * 2048 + rd0);
#pragma unroll
    for (int j = 0; j < 4; ++j) wf0[j] = *(const bf16x8*)(nbase + w_rd + j * 2048 + rd0);
    mfma16(wf1, af1);                                   // phase B (overlaps the F0 reads just issued)
    buf = nbuf;
  }
  {  // last K-step
    const char* base = lds2 + buf * GEMM2_STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) af1[i] = *(const bf16x8*)(base + a_rd + i * 2048 + rd1);
#pragma unroll
    for (int j = 0; j < 4; ++j) wf1[j] = *(const bf16x8*)(base + w_rd + j * 2048 + rd1);
    mfma16(wf0, af0);
    mfma16(wf1, af1);
  }
  gemm_epilogue(p, acc, m0 + wm * 64, n0 + wn * 64, l15, h);
}

// ---------------------------------------------------------------------------
// 256x256x64 tile, 512 threads = 8 waves (2 x 4), each wave a 128x64 output block (8 x 4 MFMA tiles, 128
// accumulator VGPRs), ONE workgroup per CU, two 64 KiB LDS buffers.  Compared with 64x64 per wave this reads
// 25 % fewer fragment bytes from LDS per MFMA and stages half the LDS-DMA bytes per FLOP.
// Per K-step t (fragments double-buffered in registers, one barrier per K-step placed between the two halves):
//   A:   ds_read F1 = frags(t, k-half 1)   || 32 MFMA on F0 = frags(t, k-half 0)
//   mid: s_waitcnt lgkmcnt(0)  - every read of buffer t%2 has landed in registers
//        s_waitcnt vmcnt(0)    - stage t+1 (issued one K-step ago) has landed
//        s_barrier             - ... for every wave; buffer t%2 is now dead
//        issue stage t+2 -> buffer t%2   (has phase B(t) + phase A(t+1) to land)
//        ds_read F0 = frags(t+1, k-half 0) from buffer (t+1)%2
//   B:   32 MFMA on F1
#define GEMM4_B 256
#define GEMM4_STAGE_BYTES (2 * GEMM4_B * GEMM_BK * 2)  // 65536
#define GEMM4_LDS_BYTES (2 * GEMM4_STAGE_BYTES)        // 131072

// NT = 16-column MFMA tiles per wave: 4 -> 256 x 256 tile, 3 -> 256 x 192 (LLM qkv: 9 x 24 = 216 tiles = one round of
// the chip instead of 1.27 rounds of 128 x 128 tiles), 2 -> 256 x 128.  Same pipeline; the W panel is NT x 64 rows.
template <int NT>
__global__ __launch_bounds__(512, 2) void gemm_bf16_256xN_kernel(GemmArgs p) {
  constexpr int BN4 = 64 * NT;                    // tile columns
  constexpr int WI = BN4 / 64;                    // W LDS-DMA instructions per thread per stage (= NT)
  constexpr int NF = 8 + NT;                      // fragment reads per k-half
  extern __shared__ __attribute__((aligned(16))) char lds4[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int l15 = lane & 15, h = lane >> 4;

  const int nwg = p.tiles_m * p.tiles_n;
  const int id = xcd_remap(blockIdx.x, nwg);
  const int tn = id / p.tiles_m, tm = id - tn * p.tiles_m;
  const int m0 = tm * GEMM4_B, n0 = tn * BN4;

  uint32_t a_off[4], w_off[WI];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = i * 512 + tid;
    const int row = c >> 3;
    const int ch = (c & 7) ^ (row & 7);
    a_off[i] = (uint32_t)(min(m0 + row, p.M - 1) - m0) * (uint32_t)(p.lda * 2) + ch * 16;
    if (i < WI) w_off[i] = (uint32_t)(min(n0 + row, p.N - 1) - n0) * (uint32_t)(p.ldw * 2) + ch * 16;
  }
  // split-K (p.part != null): grid.y slices of the K range, f32 partial tiles to part[slice][M][N]
  const int nk_all = p.K / GEMM_BK;
  const int kt0 = p.part ? (int)((long long)nk_all * blockIdx.y / p.ksplit) : 0;
  const int kt1 = p.part ? (int)((long long)nk_all * (blockIdx.y + 1) / p.ksplit) : nk_all;
  const char* a_base = (const char*)(p.A + (size_t)m0 * p.lda) + (size_t)kt0 * GEMM_BK * 2;
  const char* w_base = (const char*)(p.W + (size_t)n0 * p.ldw) + (size_t)kt0 * GEMM_BK * 2;
  const int wave_base = __builtin_amdgcn_readfirstlane(tid >> 6) * 1024;
  constexpr int A_BYTES = GEMM4_B * GEMM_BK * 2;  // 32 KiB

  // K tile kt -> LDS buffer buf.  kt is clamped by the caller, so the call is unconditional (no branch to
  // split the scheduling region); a redundant trailing stage lands in a dead buffer.
  auto stage = [&](int buf, int kt) {
    char* base = lds4 + buf * GEMM4_STAGE_BYTES + wave_base;
    const char* a_k = a_base + kt * (GEMM_BK * 2);
    const char* w_k = w_base + kt * (GEMM_BK * 2);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_k + a_off[i]),
                                       (__attribute__((address_space(3))) void*)(base + i * 8192), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < WI; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_k + w_off[i]),
                                       (__attribute__((address_space(3))) void*)(base + A_BYTES + i * 8192), 16, 0,
                                       0);
  };

  f32x4 acc[8][NT];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int sw = lane & 7;
  const int rd0 = l15 * 128 + (((0 + h) ^ sw) << 4);
  const int rd1 = l15 * 128 + (((4 + h) ^ sw) << 4);
  const int a_rd = wm * 128 * 128;            // + i * 2048
  const int w_rd = A_BYTES + wn * (16 * NT) * 128;   // + j * 2048

  bf16x8 af0[8], wf0[NT], af1[8], wf1[NT];
  auto read_frags = [&](bf16x8 (&af)[8], bf16x8 (&wf)[NT], const char* base, int rd) {
#pragma unroll
    for (int j = 0; j < NT; ++j) wf[j] = *(const bf16x8*)(base + w_rd + j * 2048 + rd);
#pragma unroll
    for (int i = 0; i < 8; ++i) af[i] = *(const bf16x8*)(base + a_rd + i * 2048 + rd);
  };
  auto mfma32 = [&](bf16x8 (&wf)[NT], bf16x8 (&af)[8], int skip_first) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
        if ((i | j) >= skip_first)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
  };

  const int nk = kt1 - kt0;
  stage(0, 0);
  stage(1, min(1, nk - 1));
  if (NT == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (NT == 3) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  read_frags(af0, wf0, lds4, rd0);
  int buf = 0;
  for (int kt = 0; kt + 1 < nk; ++kt) {
    const char* base = lds4 + buf * GEMM4_STAGE_BYTES;
    // phase A: first MFMA (the conservative lgkmcnt(0) before the first use of the back-edge-carried F0 must
    // not wait for F1), then F1 reads interleaved one per MFMA
    __builtin_amdgcn_s_setprio(1);
    acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0[0], af0[0], acc[0][0], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    read_frags(af1, wf1, base, rd1);
    mfma32(wf0, af0, 1);
#pragma unroll
    for (int g = 0; g < NF; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // one DS read
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 8 * NT - 1 - NF, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    // mid
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    stage(buf, min(kt + 2, nk - 1));
    read_frags(af0, wf0, lds4 + (buf ^ 1) * GEMM4_STAGE_BYTES, rd0);
    // phase B: MFMAs on F1 start at once; the stage issue and the F0 reads ride between them
    mfma32(wf1, af1, 0);
#pragma unroll
    for (int g = 0; g < 4 + WI; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);  // one VMEM (LDS-DMA)
    }
#pragma unroll
    for (int g = 0; g < NF; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 8 * NT - (4 + WI) - NF, 0);
    __builtin_amdgcn_sched_barrier(0);
    buf ^= 1;
  }
  {  // last K-step
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // also drains the redundant trailing stages
    const char* base = lds4 + buf * GEMM4_STAGE_BYTES;
    read_frags(af1, wf1, base, rd1);
    mfma32(wf0, af0, 0);
    mfma32(wf1, af1, 0);
  }

  if (p.part) {  // split-K: raw f32 partial sums, 16-byte stores (lane holds 4 consecutive columns)
    float* dst = p.part + (size_t)blockIdx.y * p.M * p.N;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int m = m0 + wm * 128 + i * 16 + l15;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = n0 + wn * (16 * NT) + j * 16 + 4 * h;
        if (n < p.N) *(f32x4*)(dst + (size_t)m * p.N + n) = acc[i][j];
      }
    }
    return;
  }
  if (p.wide && !(p.act == ACT_SWIGLU && (NT & 1))) {
    // every wave's fragment reads have been consumed by its MFMAs and its LDS-DMA drained (vmcnt(0) above); one
    // barrier makes that true for the whole workgroup before the buffers become staging space
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    gemm_epilogue_wide_dispatch<8, NT>(p, acc, m0 + wm * 128, n0 + wn * (16 * NT), lane, lds4 + wave * 16384);
    return;
  }
  // epilogue: two 64-row halves through the shared 4x4 epilogue
#pragma unroll
  for (int hm = 0; hm < 2; ++hm) {
    f32x4 sub[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) sub[i][j] = (j < NT) ? acc[hm * 4 + i][j < NT ? j : 0] : (f32x4){0.f, 0.f, 0.f, 0.f};
    gemm_epilogue_n(p, sub, m0 + wm * 128 + hm * 64, n0 + wn * (16 * NT), l15, h, NT);
  }
}

// ---------------------------------------------------------------------------
// 256 x 256 x 64 "ping-pong" tile: the two waves that share a SIMD (wave w and w + 4: row groups wr = 0 / 1) run
// half a phase apart, so one of them is always inside an MFMA cluster while the other issues its LDS reads and
// LDS-DMA stages.  A K-tile is four phases; a phase = {fragment reads + one half-tile stage | s_barrier |
// 16 MFMAs (one 64 x 32 quadrant of the wave's 128 x 64 output over the whole K-tile) | s_barrier}, and group
// wr = 1 enters the loop through one extra s_barrier (group 0 pays it back after the loop), which is what puts
// its read section beside group 0's MFMA section and vice versa.
//
// LDS: two K-tile buffers (E: even tiles, O: odd) of four 16 KiB half-tiles.  A half-tile is not a contiguous
// half of the tile but the rows every wave reads in the SAME phase: A-h(qa) = rows {wr*128 + qa*64 + r},
// B-h(qb) = columns {wc*64 + qb*32 + c}.  Reads per K-tile: phase 1 B-h0 (4 x ds_read_b128) + A-h0 (8),
// phase 2 B-h1 (4), phase 3 A-h1 (8), phase 4 none (quadrants (0,0) (0,1) (1,1) (1,0); B-h0 stays in registers).
// So a half-tile is dead early and is refilled for the tile two ahead while its buffer is still being read:
//   phase:   1        2        3        4        5        6        7        8
//   reads:   E.B0,A0  E.B1     E.A1     -        O.B0,A0  O.B1     O.A1     -
//   stage:   O.A1     E.B0     E.A0     E.B1     E.A1     O.B0     O.A0     O.B1      (E: tile t+2, O: t+1 / t+3)
//   wait:                               vmcnt(6)                            vmcnt(6)
// vmcnt(6) = the three youngest half-tiles stay in flight across every barrier; the wait in phase 4 (8) retires
// the odd (even) buffer, which is read from the NEXT phase on (both groups have then passed a barrier after
// their wait).  WAR: a half-tile is restaged >= 2 phases after the phase that read it, except B-h0, restaged one
// phase after - its 4 reads are issued first and retired by lgkmcnt(8) BEFORE the reading phase's first barrier.
// sched_barrier(0) at every s_barrier keeps hipcc from moving reads / MFMAs across them (the placement above IS
// the synchronisation).  Same accumulator layout and epilogue as gemm_bf16_256xN_kernel<4>.
#define GEMM5_HALF_BYTES 16384
#define GEMM5_BUF_BYTES 65536
#define GEMM5_LDS_BYTES 131072
// half-tile slots inside a buffer
#define G5_B0 0
#define G5_A0 1
#define G5_B1 2
#define G5_A1 3

// Timeline probe (tools/probes/gemm_probe.sh builds this file with -DGEMM_PROBE; never in the product build): per workgroup
// the 100 MHz clock at entry / main-loop start / main-loop end / stores issued / stores landed, and where it ran.
#ifdef GEMM_PROBE
__device__ unsigned long long g_gemm_probe[4096 * 8];
#define GP_STAMP(i) do { if (threadIdx.x == 0) g_gemm_probe[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define GP_WHERE() do { if (threadIdx.x == 0) { g_gemm_probe[blockIdx.x * 8 + 6] = __builtin_amdgcn_s_getreg((31 << 11) | 4); \
                                              g_gemm_probe[blockIdx.x * 8 + 7] = __builtin_amdgcn_s_getreg((31 << 11) | 20); } } while (0)
extern "C" int vis_gemm_probe_read(void* dst, int n_u64) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_gemm_probe), (size_t)n_u64 * 8) == hipSuccess ? 0 : 2;
}
#else
#define GP_STAMP(i) do { } while (0)
#define GP_WHERE() do { } while (0)
#endif

__global__ __launch_bounds__(512, 2) void gemm_bf16_256x256_pp_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char lds5[];
  GP_STAMP(0); GP_WHERE();
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int l15 = lane & 15, h = lane >> 4;

  const int nwg = p.tiles_m * p.tiles_n;
  const int id = xcd_remap(blockIdx.x, nwg);
  const int tn = id / p.tiles_m, tm = id - tn * p.tiles_m;
  const int m0 = tm * 256, n0 = tn * 256;

  // staging: half-tile row hr = c >> 3 (c = i*512 + tid), 16-byte chunk (c & 7) ^ (hr & 7)
  uint32_t a_off[2][2], w_off[2][2];   // [half][instruction]
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = i * 512 + tid;
    const int hr = c >> 3;
    const int ch = (c & 7) ^ (hr & 7);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int arow = (hr >> 6) * 128 + q * 64 + (hr & 63);
      const int wrow = (hr >> 5) * 64 + q * 32 + (hr & 31);
      a_off[q][i] = (uint32_t)(min(m0 + arow, p.M - 1) - m0) * (uint32_t)(p.lda * 2) + ch * 16;
      w_off[q][i] = (uint32_t)(min(n0 + wrow, p.N - 1) - n0) * (uint32_t)(p.ldw * 2) + ch * 16;
    }
  }
  const int nk_all = p.K / GEMM_BK;
  const int kt0 = p.part ? (int)((long long)nk_all * blockIdx.y / p.ksplit) : 0;
  const int kt1 = p.part ? (int)((long long)nk_all * (blockIdx.y + 1) / p.ksplit) : nk_all;
  const int nk = kt1 - kt0;
  const char* a_base = (const char*)(p.A + (size_t)m0 * p.lda) + (size_t)kt0 * GEMM_BK * 2;
  const char* w_base = (const char*)(p.W + (size_t)n0 * p.ldw) + (size_t)kt0 * GEMM_BK * 2;
  char* const wave_lds = lds5 + wave * 1024;

  // one half-tile (slot of buffer buf) of K-tile kt; kt is clamped, a redundant stage refills a dead half-tile
  auto stage = [&](int buf, int slot, int kt) {
    kt = min(kt, nk - 1);
    const bool is_a = slot & 1;
    const int q = slot >> 1;
    const char* src = (is_a ? a_base : w_base) + kt * (GEMM_BK * 2);
    char* dst = wave_lds + buf * GEMM5_BUF_BYTES + slot * GEMM5_HALF_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const __attribute__((address_space(1))) void* g =
          (const __attribute__((address_space(1))) void*)(src + (is_a ? a_off[q][i] : w_off[q][i]));
      __attribute__((address_space(3))) void* l = (__attribute__((address_space(3))) void*)(dst + i * 8192);
      __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
    }
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment reads: half-tile row l15 (+ 16 per fragment), k-half kh -> chunk (4 kh + h) ^ (row & 7)
  const int sw = l15 & 7;
  const int rd_k0 = l15 * 128 + (((0 + h) ^ sw) << 4);
  const int rd_k1 = l15 * 128 + (((4 + h) ^ sw) << 4);
  const int a_rd = wr * (64 * 128);   // + ii * 2048
  const int b_rd = wc * (32 * 128);   // + jj * 2048
  bf16x8 af[4][2], bq0[2][2], bq1[2][2];
  auto read_a = [&](int buf, int qa) {
    const char* base = lds5 + buf * GEMM5_BUF_BYTES + (qa ? G5_A1 : G5_A0) * GEMM5_HALF_BYTES + a_rd;
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
      af[ii][0] = *(const bf16x8*)(base + ii * 2048 + rd_k0);
      af[ii][1] = *(const bf16x8*)(base + ii * 2048 + rd_k1);
    }
  };
  auto read_b = [&](bf16x8 (&bq)[2][2], int buf, int qb) {
    const char* base = lds5 + buf * GEMM5_BUF_BYTES + (qb ? G5_B1 : G5_B0) * GEMM5_HALF_BYTES + b_rd;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      bq[jj][0] = *(const bf16x8*)(base + jj * 2048 + rd_k0);
      bq[jj][1] = *(const bf16x8*)(base + jj * 2048 + rd_k1);
    }
  };
#define G5_BAR()                                  \
  do {                                            \
    __builtin_amdgcn_sched_barrier(0);            \
    asm volatile("s_barrier" ::: "memory");       \
    __builtin_amdgcn_sched_barrier(0);            \
  } while (0)
#define G5_MFMA(QA, QB, BQ)                                                                                   \
  do {                                                                                                        \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                        \
    __builtin_amdgcn_s_setprio(1);                                                                            \
    _Pragma("unroll") for (int kh = 0; kh < 2; ++kh)                                                          \
    _Pragma("unroll") for (int ii = 0; ii < 4; ++ii)                                                          \
    _Pragma("unroll") for (int jj = 0; jj < 2; ++jj)                                                          \
      acc[(QA) * 4 + ii][(QB) * 2 + jj] =                                                                     \
          __builtin_amdgcn_mfma_f32_16x16x32_bf16(BQ[jj][kh], af[ii][kh], acc[(QA) * 4 + ii][(QB) * 2 + jj], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                            \
  } while (0)
  // four phases of K-tile kt in buffer B (0 = E, 1 = O); S1..S4 = (buffer, slot, tile) staged in each phase
#define G5_TILE(B, KT, S1B, S1S, S1T, S2B, S2S, S2T, S3B, S3S, S3T, S4B, S4S, S4T)  \
  do {                                                                              \
    read_b(bq0, B, 0);                                                              \
    __builtin_amdgcn_sched_barrier(0);                                              \
    read_a(B, 0);                                                                   \
    stage(S1B, S1S, S1T);                                                           \
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");                              \
    G5_BAR();                                                                       \
    G5_MFMA(0, 0, bq0);                                                             \
    G5_BAR();                                                                       \
    read_b(bq1, B, 1);                                                              \
    stage(S2B, S2S, S2T);                                                           \
    G5_BAR();                                                                       \
    G5_MFMA(0, 1, bq1);                                                             \
    G5_BAR();                                                                       \
    read_a(B, 1);                                                                   \
    stage(S3B, S3S, S3T);                                                           \
    G5_BAR();                                                                       \
    G5_MFMA(1, 1, bq1);                                                             \
    G5_BAR();                                                                       \
    stage(S4B, S4S, S4T);                                                           \
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                \
    G5_BAR();                                                                       \
    G5_MFMA(1, 0, bq0);                                                             \
    G5_BAR();                                                                       \
  } while (0)

  // prologue: tile 0 complete (E), tile 1 less its A-h1 (O): 7 half-tiles; vmcnt(6) = E has landed
  stage(0, G5_B0, 0); stage(0, G5_A0, 0); stage(0, G5_B1, 0); stage(0, G5_A1, 0);
  stage(1, G5_B0, 1); stage(1, G5_A0, 1); stage(1, G5_B1, 1);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  G5_BAR();
  if (wr == 1) G5_BAR();   // stagger: group 1 runs one barrier behind group 0

  GP_STAMP(1);
  int t = 0;
  for (; t + 1 < nk; t += 2) {
    G5_TILE(0, t, 1, G5_A1, t + 1, 0, G5_B0, t + 2, 0, G5_A0, t + 2, 0, G5_B1, t + 2);
    G5_TILE(1, t + 1, 0, G5_A1, t + 2, 1, G5_B0, t + 3, 1, G5_A0, t + 3, 1, G5_B1, t + 3);
  }
  if (t < nk) G5_TILE(0, t, 1, G5_A1, t + 1, 0, G5_B0, t + 2, 0, G5_A0, t + 2, 0, G5_B1, t + 2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // redundant trailing stages
  if (wr == 0) G5_BAR();   // pay the stagger back: every wave has now executed the same number of barriers
  GP_STAMP(2);
#undef G5_TILE
#undef G5_MFMA
#undef G5_BAR

  if (p.part) {  // split-K: raw f32 partial sums
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
    GP_STAMP(3);
#ifdef GEMM_PROBE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GP_STAMP(4);
#endif
    return;
  }
  if (p.wide) {
    // all waves have executed the same number of barriers and drained their own LDS-DMA (vmcnt(0) above), but a
    // faster wave must not overwrite a half-tile a trailing redundant stage of ANOTHER wave is still landing in, nor
    // fragments a slower wave has yet to read: one more barrier for everyone, then the buffers are staging space
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#if defined(GEMM_PROBE) && GEMM_PROBE == 2   // every workgroup stores to tile (0, 0): no HBM write traffic
    gemm_epilogue_wide_dispatch<8, 4>(p, acc, wr * 128, wc * 64, lane, lds5 + wave * 16384);
#else
    gemm_epilogue_wide_dispatch<8, 4>(p, acc, m0 + wr * 128, n0 + wc * 64, lane, lds5 + wave * 16384);
#endif
    GP_STAMP(3);
#ifdef GEMM_PROBE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GP_STAMP(4);
#endif
    return;
  }
#pragma unroll
  for (int hm = 0; hm < 2; ++hm) {
    f32x4 sub[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) sub[i][j] = acc[hm * 4 + i][j];
    gemm_epilogue_n(p, sub, m0 + wr * 128 + hm * 64, n0 + wc * 64, l15, h, 4);
  }
}

// ---------------------------------------------------------------------------
// 128 x 256 x 64 ping-pong tile: the half-height form of the kernel above, for grids whose 256 x 256 tiling is ragged
// (LLM o: 9 x 14 = 126 tiles on 256 CUs; ViT proj: 20 x 5 = 100; the gate/up remainder columns) and which so far ran on
// the 128 x 128 kernel at a quarter of the ping-pong tile's rate.  Same eight waves, same two-groups-half-a-phase-apart
// pairing, same MFMA, operand order and K order (results are bit-identical to every other bf16 GEMM kernel here); a wave
// owns 64 x 64 of the output, so a K-tile is TWO phases:
//   phase 1: reads B-h0 (4 x ds_read_b128) + A (8) | s_barrier | 16 MFMAs (columns 0..31 of the wave) | s_barrier
//   phase 2: reads B-h1 (4)                        | s_barrier | 16 MFMAs (columns 32..63)            | s_barrier
// LDS: THREE K-tile buffers of three 16 KiB slots (B-h0, A, B-h1 = 144 KiB); tile t + 2 is staged while tile t is
// consumed - B-h0 and A in phase 1, B-h1 in phase 2, each into the slot of tile t - 1 that was last read two phases
// earlier (the WAR distance the kernel above needs between a read and the restage of its slot, with the groups half a
// phase apart).  One counted wait per K-tile: vmcnt(6) at the end of phase 2 of tile t leaves only the six LDS-DMA
// instructions of tile t + 2 in flight, so tile t + 1 is complete before its first read (next phase, after the barrier).
// Bytes per flop are 1.5 x the 256 x 256 tile's (48 KiB of operands per K-tile for half the MFMAs), so this kernel sits
// nearer the CU's LDS-DMA rate than the MFMA pipe; it is chosen only where whole rounds of 256 x 256 tiles do not exist.
#define GEMM6_SLOT_BYTES 16384
#define GEMM6_BUF_BYTES 49152
#define GEMM6_LDS_BYTES 147456
#define G6_B0 0
#define G6_A 1
#define G6_B1 2

__global__ __launch_bounds__(512, 2) void gemm_bf16_128x256_pp_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char lds6[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int l15 = lane & 15, h = lane >> 4;

  const int nwg = p.tiles_m * p.tiles_n;
  const int id = xcd_remap(blockIdx.x, nwg);
  const int tn = id / p.tiles_m, tm = id - tn * p.tiles_m;
  const int m0 = tm * 128, n0 = tn * 256;

  // staging: slot row hr = c >> 3 (c = i*512 + tid), 16-byte chunk (c & 7) ^ (hr & 7); A slot row = tile row,
  // B-h(q) slot row hr = column (hr >> 5) * 64 + q * 32 + (hr & 31) (the columns every wave reads in phase q + 1)
  uint32_t a_off[2], w_off[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = i * 512 + tid;
    const int hr = c >> 3;
    const int ch = (c & 7) ^ (hr & 7);
    a_off[i] = (uint32_t)(min(m0 + hr, p.M - 1) - m0) * (uint32_t)(p.lda * 2) + ch * 16;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int wrow = (hr >> 5) * 64 + q * 32 + (hr & 31);
      w_off[q][i] = (uint32_t)(min(n0 + wrow, p.N - 1) - n0) * (uint32_t)(p.ldw * 2) + ch * 16;
    }
  }
  const int nk = p.K / GEMM_BK;
  const char* a_base = (const char*)(p.A + (size_t)m0 * p.lda);
  const char* w_base = (const char*)(p.W + (size_t)n0 * p.ldw);
  char* const wave_lds = lds6 + wave * 1024;

  // one slot of buffer buf for K-tile kt; kt is clamped, a redundant stage refills a dead slot
  auto stage = [&](int buf, int slot, int kt) {
    kt = min(kt, nk - 1);
    const char* src = (slot == G6_A ? a_base : w_base) + kt * (GEMM_BK * 2);
    char* dst = wave_lds + buf * GEMM6_BUF_BYTES + slot * GEMM6_SLOT_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const uint32_t off = (slot == G6_A) ? a_off[i] : (slot == G6_B0 ? w_off[0][i] : w_off[1][i]);
      const __attribute__((address_space(1))) void* g = (const __attribute__((address_space(1))) void*)(src + off);
      __attribute__((address_space(3))) void* l = (__attribute__((address_space(3))) void*)(dst + i * 8192);
      __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int sw = l15 & 7;
  const int rd_k0 = l15 * 128 + (((0 + h) ^ sw) << 4);
  const int rd_k1 = l15 * 128 + (((4 + h) ^ sw) << 4);
  const int a_rd = wr * (64 * 128);   // + ii * 2048
  const int b_rd = wc * (32 * 128);   // + jj * 2048
  bf16x8 af[4][2], bq0[2][2], bq1[2][2];
  auto read_a = [&](int buf) {
    const char* base = lds6 + buf * GEMM6_BUF_BYTES + G6_A * GEMM6_SLOT_BYTES + a_rd;
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
      af[ii][0] = *(const bf16x8*)(base + ii * 2048 + rd_k0);
      af[ii][1] = *(const bf16x8*)(base + ii * 2048 + rd_k1);
    }
  };
  auto read_b = [&](bf16x8 (&bq)[2][2], int buf, int qb) {
    const char* base = lds6 + buf * GEMM6_BUF_BYTES + (qb ? G6_B1 : G6_B0) * GEMM6_SLOT_BYTES + b_rd;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      bq[jj][0] = *(const bf16x8*)(base + jj * 2048 + rd_k0);
      bq[jj][1] = *(const bf16x8*)(base + jj * 2048 + rd_k1);
    }
  };
#define G6_BAR()                                  \
  do {                                            \
    __builtin_amdgcn_sched_barrier(0);            \
    asm volatile("s_barrier" ::: "memory");       \
    __builtin_amdgcn_sched_barrier(0);            \
  } while (0)
#define G6_MFMA(QB, BQ)                                                                                       \
  do {                                                                                                        \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                        \
    __builtin_amdgcn_s_setprio(1);                                                                            \
    _Pragma("unroll") for (int kh = 0; kh < 2; ++kh)                                                          \
    _Pragma("unroll") for (int ii = 0; ii < 4; ++ii)                                                          \
    _Pragma("unroll") for (int jj = 0; jj < 2; ++jj)                                                          \
      acc[ii][(QB) * 2 + jj] =                                                                                \
          __builtin_amdgcn_mfma_f32_16x16x32_bf16(BQ[jj][kh], af[ii][kh], acc[ii][(QB) * 2 + jj], 0, 0, 0);   \
    __builtin_amdgcn_s_setprio(0);                                                                            \
  } while (0)
  // the two phases of K-tile T in buffer B; tile T + 2 goes to buffer NB = (B + 2) % 3
#define G6_TILE(B, NB, T)                                 \
  do {                                                    \
    read_b(bq0, B, 0);                                    \
    __builtin_amdgcn_sched_barrier(0);                    \
    read_a(B);                                            \
    stage(NB, G6_B0, (T) + 2);                            \
    stage(NB, G6_A, (T) + 2);                             \
    G6_BAR();                                             \
    G6_MFMA(0, bq0);                                      \
    G6_BAR();                                             \
    read_b(bq1, B, 1);                                    \
    stage(NB, G6_B1, (T) + 2);                            \
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      \
    G6_BAR();                                             \
    G6_MFMA(1, bq1);                                      \
    G6_BAR();                                             \
  } while (0)

  // prologue: tiles 0 and 1 complete (buffers 0, 1): 12 instructions, vmcnt(6) = tile 0 has landed
  stage(0, G6_B0, 0); stage(0, G6_A, 0); stage(0, G6_B1, 0);
  stage(1, G6_B0, 1); stage(1, G6_A, 1); stage(1, G6_B1, 1);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  G6_BAR();
  if (wr == 1) G6_BAR();   // stagger: group 1 runs one barrier behind group 0

  int t = 0;
  for (; t + 2 < nk; t += 3) {
    G6_TILE(0, 2, t);
    G6_TILE(1, 0, t + 1);
    G6_TILE(2, 1, t + 2);
  }
  if (t < nk) {
    G6_TILE(0, 2, t);
    if (t + 1 < nk) G6_TILE(1, 0, t + 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // redundant trailing stages
  if (wr == 0) G6_BAR();   // pay the stagger back: every wave has now executed the same number of barriers
#undef G6_TILE
#undef G6_MFMA
#undef G6_BAR

  if (p.wide) {
    // as in the kernel above: one more barrier for everyone, then the buffers are staging space
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    gemm_epilogue_wide_dispatch<4, 4>(p, acc, m0 + wr * 64, n0 + wc * 64, lane, lds6 + wave * 16384);
    return;
  }
  gemm_epilogue_n(p, acc, m0 + wr * 64, n0 + wc * 64, l15, h, 4);
}

// ---------------------------------------------------------------------------
// Batched decode projection (M = in-flight sequences <= 16): a pure weight-streaming problem, so the kernel is
// built around BYTES IN FLIGHT, not around the MFMA.
//  * The (128-column tile, 64-wide K-step) pairs of the whole projection form ONE sequence of
//    T = tiles x K/64 steps, cut into <= 256 equal contiguous ranges - one persistent workgroup per CU
//    (LDS-limited on purpose), each streaming ~T/256 x 16 KiB of W without a gap at tile seams ("stream-K").
//  * A ring of 7 stages (7 x (16 KiB of W + 4 KiB of x)) is filled by LDS-DMA in full 128-B lines (a
//    lane-per-row "fragment shaped" global load is texture-addresser bound at ~2.5 TB/s); up to 6 x 16 KiB of
//    weights are outstanding per CU, i.e. ~4 us of latency tolerance at the chip's 6 TB/s.  Earlier versions
//    (one stage in flight per workgroup; then a ring per (tile, K-slice) workgroup whose fill/drain bubble
//    was paid once per short slice) streamed at 2.3 - 2.9 TB/s.
//  * Per step: counted s_waitcnt vmcnt (the oldest stage has landed, 5 younger ones stay in flight; stores
//    of a flush can only make the wait conservative because loads retire in order), ONE s_barrier, refill
//    of the slot consumed one step ago, 6 ds_read_b128 + 4 MFMAs per wave.
//  * A workgroup that leaves a tile writes its f32 partial to part[seg][16][N], seg = its rank among the
//    workgroups touching that tile (<= nslots, fixed by the geometry: results are bitwise reproducible);
//    the one that finishes the tile zero-fills the unused slots, so vis_skinny_finalize sums a fixed
//    number of slots.  With part == NULL ranges are whole tiles and C is written directly (bf16 or f32 logits).
#define GEMM3_BM 32
#define GEMM3_STAGE_BYTES ((GEMM3_BM + GEMM_BN) * GEMM_BK * 2)  // 20480
#define GEMM3_DEPTH 7
#define GEMM3_LDS_BYTES (GEMM3_DEPTH * GEMM3_STAGE_BYTES)      // 143360
#define GEMM3_PER 5                                             // LDS-DMA instructions per thread per stage
#define GEMM3_MAX_WG 256                                        // one per CU
#define GEMM3_MAX_SLOTS 16

struct DecGemmArgs {
  const bf16_t* A;  // [M][lda]
  const bf16_t* W;  // [N][ldw]
  float* part;      // [nslots][16][N] or null
  void* C;          // direct output (part == null)
  int M, N, lda, ldw, ldc;   // lda / ldw in ELEMENTS (= bytes for fp8)
  int nk_all, total, spb, nslots, out_f32;
  // fp8 form (template FP8): A / W are OCP e4m3 bytes, a K-step is 128 elements (again 128 bytes per tile row);
  // the direct-output path applies the row scales sx[m] * sw[n]; partials stay raw (vis_skinny_finalize scales them)
  const float* sx;
  const float* sw;
};

__device__ __forceinline__ void gemm3_wait_stages(int younger) {
  // wait until all but `younger` (0..DEPTH-2) most recent stages of this wave have landed
  switch (younger) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(25)" ::: "memory"); break;
  }
}
static_assert(GEMM3_DEPTH == 7 && GEMM3_PER == 5, "gemm3_wait_stages encodes (DEPTH-2) x PER");
// MB = 4 (33..64 sequences): the x tile is 64 rows (8 KiB), a stage 24 KiB, six LDS-DMA instructions per thread per
// stage, and the ring is 6 deep (6 x 24 KiB = 144 KiB)
#ifndef GEMM3W_DEPTH
#define GEMM3W_DEPTH 6
#endif
#define GEMM3W_STAGE_BYTES ((64 + GEMM_BN) * GEMM_BK * 2)        // 24576
#define GEMM3W_LDS_BYTES (GEMM3W_DEPTH * GEMM3W_STAGE_BYTES)     // 147456
__device__ __forceinline__ void gemm3w_wait_stages(int younger) {
  switch (younger) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
  }
}

// MB = 16-row blocks of x: 1 for M <= 16 sequences, 2 for M <= 32 (x tile of 32 rows, 7-deep ring), 4 for M <= 64 (x tile
// of 64 rows, 6-deep ring); the partial slab of a slot has 16 * MB rows.
template <bool FP8, int MB>
__global__ __launch_bounds__(256, 1) void gemm_decode_stream_kernel(DecGemmArgs p) {
  constexpr int EB = FP8 ? 1 : 2;  // bytes per element
  extern __shared__ __attribute__((aligned(16))) char lds3[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wn = tid >> 6;  // wave = 32-column slice
  const int l15 = lane & 15, h = lane >> 4;
  const int s0 = blockIdx.x * p.spb;
  const int nsteps = min(s0 + p.spb, p.total) - s0;
  if (nsteps <= 0) return;  // whole workgroup

  // staging slots: x tile 32 (MB = 4: 64) rows x 8 chunks (1 or 2 per thread; rows >= M repeat row M-1), W tile 128 x 8
  // (4 per thread)
  constexpr int XI = (MB == 4) ? 2 : 1;                 // x LDS-DMA instructions per thread per stage
  constexpr int DEPTH = (MB == 4) ? GEMM3W_DEPTH : GEMM3_DEPTH;
  constexpr int STAGE_BYTES = (MB == 4) ? GEMM3W_STAGE_BYTES : GEMM3_STAGE_BYTES;
  uint32_t a_off[XI], w_off[4];
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int row = i * 32 + (tid >> 3), ch = (tid & 7) ^ (row & 7);
    a_off[i] = (uint32_t)min(row, p.M - 1) * (uint32_t)(p.lda * EB) + ch * 16;
  }
  int p_tile = s0 / p.nk_all, p_kt = s0 - p_tile * p.nk_all;  // producer cursor
  const char* a_ptr;
  const char* w_ptr;
  auto enter_tile = [&](int tile, int kt) {
    const int n0 = tile * GEMM_BN;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = i * 256 + tid;
      const int row = c >> 3, ch = (c & 7) ^ (row & 7);
      w_off[i] = (uint32_t)(min(n0 + row, p.N - 1) - n0) * (uint32_t)(p.ldw * EB) + ch * 16;
    }
    a_ptr = (const char*)p.A + (size_t)kt * 128;
    w_ptr = (const char*)p.W + (size_t)n0 * p.ldw * EB + (size_t)kt * 128;
  };
  enter_tile(p_tile, p_kt);
  const int wave_base = __builtin_amdgcn_readfirstlane(tid >> 6) * 1024;
  constexpr int A_BYTES = XI * GEMM3_BM * GEMM_BK * 2;  // 4 KiB (MB = 4: 8 KiB)

  auto stage = [&](int slot) {  // next step of this workgroup's range -> ring slot
    char* base = lds3 + slot * STAGE_BYTES + wave_base;
#pragma unroll
    for (int i = 0; i < XI; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_ptr + a_off[i]),
                                       (__attribute__((address_space(3))) void*)(base + i * 4096), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_ptr + w_off[i]),
                                       (__attribute__((address_space(3))) void*)(base + A_BYTES + i * 4096), 16, 0, 0);
    a_ptr += 128;
    w_ptr += 128;
    if (++p_kt == p.nk_all) {
      p_kt = 0;
      ++p_tile;
      enter_tile(p_tile, 0);  // never dereferenced past the last tile: the caller stops issuing at nsteps
    }
  };

  const int sw = lane & 7;
  const int rd0 = l15 * 128 + (((0 + h) ^ sw) << 4);
  const int rd1 = l15 * 128 + (((4 + h) ^ sw) << 4);
  const int w_rd = A_BYTES + wn * 32 * 128;

  int c_tile = s0 / p.nk_all, c_kt = s0 - c_tile * p.nk_all;  // consumer cursor
  const int pre = min(DEPTH - 1, nsteps);
  for (int s = 0; s < pre; ++s) stage(s);
  int slot = 0, fill = pre % DEPTH;  // slot consumed this step / slot refilled this step
  int st = 0;

  // one K-step of the ring into `acc`
  auto step = [&](f32x4 (&acc)[MB][2]) {
    if constexpr (MB == 4) gemm3w_wait_stages(min(DEPTH - 2, nsteps - 1 - st));
    else gemm3_wait_stages(min(DEPTH - 2, nsteps - 1 - st));
    __builtin_amdgcn_s_barrier();  // stage st visible to all waves; every wave is past compute(st-1)
    if (st + DEPTH - 1 < nsteps) {
      stage(fill);
      fill = (fill + 1 == DEPTH) ? 0 : fill + 1;
    }
    const char* base = lds3 + slot * STAGE_BYTES;
    if constexpr (FP8) {
      // lane holds row l15, k = 32 h .. 32 h + 31: 16-byte chunks 2h and 2h+1 at their swizzled positions
      typedef int i32x8 __attribute__((ext_vector_type(8)));
      const int qlo = l15 * 128 + (((2 * h) ^ sw) << 4), qhi = l15 * 128 + (((2 * h + 1) ^ sw) << 4);
      auto frag8 = [&](const char* p0) -> i32x8 {
        const u32x4 lo = *(const u32x4*)(p0 + qlo), hi = *(const u32x4*)(p0 + qhi);
        return (i32x8){(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
      };
      i32x8 xa[MB];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) xa[mb] = frag8(base + mb * 2048);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const i32x8 wa = frag8(base + w_rd + j * 2048);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
          acc[mb][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wa, xa[mb], acc[mb][j], 0, 0, 0, 0x7f7f7f7f, 0,
                                                                        0x7f7f7f7f);
      }
    } else {
      bf16x8 a0[MB], a1[MB];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        a0[mb] = *(const bf16x8*)(base + mb * 2048 + rd0);
        a1[mb] = *(const bf16x8*)(base + mb * 2048 + rd1);
      }
      bf16x8 wf[2][2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        wf[j][0] = *(const bf16x8*)(base + w_rd + j * 2048 + rd0);
        wf[j][1] = *(const bf16x8*)(base + w_rd + j * 2048 + rd1);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          acc[mb][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][0], a0[mb], acc[mb][j], 0, 0, 0);
          acc[mb][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][1], a1[mb], acc[mb][j], 0, 0, 0);
        }
    }
    slot = (slot + 1 == DEPTH) ? 0 : slot + 1;
    ++st;
  };

  // flush of one tile segment: lane holds D[n = n0 + 32 wn + 16 j + 4 h + r][m = 16 mb + l15]
  auto flush = [&](f32x4 (&acc)[MB][2], int tile, bool tile_done) {
    const int nb = tile * GEMM_BN + wn * 32 + 4 * h;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const int m = mb * 16 + l15;
      if (m >= p.M) continue;
      if (p.part) {
        const int seg = blockIdx.x - (tile * p.nk_all) / p.spb;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int n = nb + j * 16;
          if (n >= p.N) continue;
          *(f32x4*)(p.part + ((size_t)seg * (16 * MB) + m) * p.N + n) = acc[mb][j];
          if (tile_done)
            for (int z = seg + 1; z < p.nslots; ++z)
              *(f32x4*)(p.part + ((size_t)z * (16 * MB) + m) * p.N + n) = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int n = nb + j * 16;
          if (n >= p.N) continue;
          f32x4 v = acc[mb][j];
          if (FP8) {   // direct output of the fp8 form: apply the row scales here
            const float sxm = p.sx[m];
            const f32x4 s4 = *(const f32x4*)(p.sw + n);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= sxm * s4[r];
          }
          if (p.out_f32) {
            *(f32x4*)((float*)p.C + (size_t)m * p.ldc + n) = v;
          } else {
            u32x2 o;
            o[0] = pack2bf(v[0], v[1]);
            o[1] = pack2bf(v[2], v[3]);
            *(u32x2*)((bf16_t*)p.C + (size_t)m * p.ldc + n) = o;
          }
        }
      }
    }
  };

  // A workgroup's range touches a few tiles (gate/up: 65 steps over 56-step tiles = up to three).  Each gets its OWN
  // accumulator set and ALL sets are stored after the last K-step: a store issued while the ring is running sits in the
  // same in-order-counted vmcnt queue as the ring's LDS-DMA loads, and the counted waits of the following steps then
  // wait for it too - probe builds without the stores ran 20-23 % faster at 64 rows (gate/up 74 -> 57 us, lm_head
  // 250 -> 196), with the same stores aimed at an L2-resident scratch not at all.  Ranges that touch more than NSEG
  // tiles (small K: many short tiles) flush the oldest set on the spot, as before.
  constexpr int NSEG = (MB == 4) ? 6 : 8;
  f32x4 accs[NSEG][MB][2];
  int seg_tile[NSEG];
  bool seg_done[NSEG];
  int nseg = 0;
#pragma unroll
  for (int sg = 0; sg < NSEG; ++sg) {
    seg_tile[sg] = 0;
    seg_done[sg] = false;
    if (st < nsteps) {   // workgroup-uniform
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        accs[sg][mb][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
        accs[sg][mb][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      for (;;) {
        const int n_here = min(nsteps - st, p.nk_all - c_kt);
        for (int i = 0; i < n_here; ++i) step(accs[sg]);
        c_kt += n_here;
        const bool done = (c_kt == p.nk_all);
        seg_tile[sg] = c_tile;
        seg_done[sg] = done;
        if (done) { c_kt = 0; ++c_tile; }
        if (sg + 1 < NSEG || st >= nsteps) break;
        // last set and steps left: this tile is stored now and the set reused
        flush(accs[sg], seg_tile[sg], seg_done[sg]);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          accs[sg][mb][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
          accs[sg][mb][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      }
      nseg = sg + 1;
    }
  }
#pragma unroll
  for (int sg = 0; sg < NSEG; ++sg)
    if (sg < nseg) flush(accs[sg], seg_tile[sg], seg_done[sg]);
}

// geometry of the stream-K cut for (N, K): steps per workgroup and the number of partial slots a tile can need
static void gemm3_geometry(int N, int K, bool direct, int* spb, int* nslots, int* nwg, int kstep = GEMM_BK) {
  const int tiles = (N + GEMM_BN - 1) / GEMM_BN, nk = K / kstep;
  const long total = (long)tiles * nk;
  int wg = GEMM3_MAX_WG;
  for (;;) {
    long per;
    if (direct) {
      const long tpw = (tiles + wg - 1) / wg;  // whole tiles per workgroup
      per = tpw * nk;
    } else {
      per = (total + wg - 1) / wg;
      if (per < 4) per = total < 4 ? total : 4;  // never cut finer than 4 K-steps
    }
    int slots = 1;
    if (!direct)
      for (int t = 0; t < tiles; ++t) {
        const int s = (int)(((long)(t + 1) * nk - 1) / per - ((long)t * nk) / per) + 1;
        if (s > slots) slots = s;
      }
    if (slots <= GEMM3_MAX_SLOTS || wg == 1) {
      *spb = (int)per;
      *nslots = slots;
      *nwg = (int)((total + per - 1) / per);
      return;
    }
    wg = wg / 2;  // fewer, longer ranges -> fewer segments per tile
  }
}

// Batched-decode projection, first half: part[slot][16][N] (f32, `ksplit` slots, all written) with
// sum_slot part = x[B,K] * W[N,K]^T, or with part == NULL a direct C (bf16, or f32 when out_f32).  B <= 64; a slot's
// slab has 16 rows for B <= 16, 32 rows for B <= 32 and 64 rows beyond (vis_skinny_finalize applies the same rule).
// ksplit <= 0 means vis_gemm_decode_ksplit(N, K); a larger value only adds zero-filled slots.
extern "C" int vis_gemm_decode_bf16(const void* A, const void* W, void* part, void* C, int B, int N, int K, int lda,
                                    int ldw, int ldc, int ksplit, int out_f32, hipStream_t stream) {
  if (!A || !W || (!part && !C) || B <= 0 || B > 64 || N <= 0 || K <= 0) return VIS_ERR_ARG;
  if (K % GEMM_BK != 0 || N % 4 != 0 || lda % 8 != 0 || ldw % 8 != 0 || (C && ldc % 4 != 0)) return VIS_ERR_ARG;
  if (((uintptr_t)A | (uintptr_t)W | (uintptr_t)part | (uintptr_t)C) & 15) return VIS_ERR_ARG;
  int spb, need, nwg;
  gemm3_geometry(N, K, part == nullptr, &spb, &need, &nwg);
  if (part) {
    if (ksplit <= 0) ksplit = need;
    if (ksplit < need || ksplit > GEMM3_MAX_SLOTS) return VIS_ERR_ARG;
  } else {
    ksplit = 1;
  }
  static const bool attr3_ok = [] {
    return hipFuncSetAttribute((const void*)gemm_decode_stream_kernel<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               GEMM3_LDS_BYTES) == hipSuccess &&
           hipFuncSetAttribute((const void*)gemm_decode_stream_kernel<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               GEMM3_LDS_BYTES) == hipSuccess &&
           hipFuncSetAttribute((const void*)gemm_decode_stream_kernel<false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               GEMM3W_LDS_BYTES) == hipSuccess;
  }();
  if (!attr3_ok) return VIS_ERR_LAUNCH;
  DecGemmArgs p;
  p.A = (const bf16_t*)A; p.W = (const bf16_t*)W; p.part = (float*)part; p.C = C;
  p.M = B; p.N = N; p.lda = lda; p.ldw = ldw; p.ldc = ldc;
  p.nk_all = K / GEMM_BK;
  p.total = ((N + GEMM_BN - 1) / GEMM_BN) * p.nk_all;
  p.spb = spb; p.nslots = ksplit; p.out_f32 = out_f32; p.sx = nullptr; p.sw = nullptr;
  vis_clear_error();
  if (B > 32) hipLaunchKernelGGL((gemm_decode_stream_kernel<false, 4>), dim3(nwg), dim3(256), GEMM3W_LDS_BYTES, stream, p);
  else if (B > 16) hipLaunchKernelGGL((gemm_decode_stream_kernel<false, 2>), dim3(nwg), dim3(256), GEMM3_LDS_BYTES, stream, p);
  else hipLaunchKernelGGL((gemm_decode_stream_kernel<false, 1>), dim3(nwg), dim3(256), GEMM3_LDS_BYTES, stream, p);
  return vis_check_launch();
}

// number of partial slots vis_gemm_decode_bf16 writes for (N, K): size part as slots x (B <= 16 ? 16 : B <= 32 ? 32 : 64) x N floats
extern "C" int vis_gemm_decode_ksplit(int N, int K) {
  if (N <= 0 || K < GEMM_BK) return 0;
  int spb, slots, nwg;
  gemm3_geometry(N, K, false, &spb, &slots, &nwg);
  return slots;
}

// fp8 form of the batched decode projection (BASELINE configs[4]): xq [B][ldx] / Wq [N][ldw] OCP e4m3 bytes with row
// scales sx [B] / sw [N].  Partials (part != NULL) are RAW sums - vis_skinny_finalize applies sx * sw; the direct
// form (part == NULL) scales in the kernel.  K % 128 == 0.
extern "C" int vis_gemm_decode_fp8(const void* xq, const void* sx, const void* Wq, const void* sw, void* part, void* C,
                                   int B, int N, int K, int ldx, int ldw, int ldc, int ksplit, int out_f32,
                                   hipStream_t stream) {
  if (!xq || !Wq || (!part && !C) || B <= 0 || B > 64 || N <= 0 || K <= 0) return VIS_ERR_ARG;
  if (K % 128 != 0 || N % 4 != 0 || ldx % 16 != 0 || ldw % 16 != 0 || ldx < K || ldw < K || (C && ldc % 4 != 0))
    return VIS_ERR_ARG;
  if (!part && (!sx || !sw)) return VIS_ERR_ARG;
  if (((uintptr_t)xq | (uintptr_t)Wq | (uintptr_t)part | (uintptr_t)C | (uintptr_t)sw) & 15) return VIS_ERR_ARG;
  int spb, need, nwg;
  gemm3_geometry(N, K, part == nullptr, &spb, &need, &nwg, 128);
  if (part) {
    if (ksplit <= 0) ksplit = need;
    if (ksplit < need || ksplit > GEMM3_MAX_SLOTS) return VIS_ERR_ARG;
  } else {
    ksplit = 1;
  }
  static const bool attr_ok = [] {
    return hipFuncSetAttribute((const void*)gemm_decode_stream_kernel<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               GEMM3_LDS_BYTES) == hipSuccess &&
           hipFuncSetAttribute((const void*)gemm_decode_stream_kernel<true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               GEMM3_LDS_BYTES) == hipSuccess &&
           hipFuncSetAttribute((const void*)gemm_decode_stream_kernel<true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               GEMM3W_LDS_BYTES) == hipSuccess;
  }();
  if (!attr_ok) return VIS_ERR_LAUNCH;
  DecGemmArgs p;
  p.A = (const bf16_t*)xq; p.W = (const bf16_t*)Wq; p.part = (float*)part; p.C = C;
  p.M = B; p.N = N; p.lda = ldx; p.ldw = ldw; p.ldc = ldc;
  p.nk_all = K / 128;
  p.total = ((N + GEMM_BN - 1) / GEMM_BN) * p.nk_all;
  p.spb = spb; p.nslots = ksplit; p.out_f32 = out_f32; p.sx = (const float*)sx; p.sw = (const float*)sw;
  vis_clear_error();
  if (B > 32) hipLaunchKernelGGL((gemm_decode_stream_kernel<true, 4>), dim3(nwg), dim3(256), GEMM3W_LDS_BYTES, stream, p);
  else if (B > 16) hipLaunchKernelGGL((gemm_decode_stream_kernel<true, 2>), dim3(nwg), dim3(256), GEMM3_LDS_BYTES, stream, p);
  else hipLaunchKernelGGL((gemm_decode_stream_kernel<true, 1>), dim3(nwg), dim3(256), GEMM3_LDS_BYTES, stream, p);
  return vis_check_launch();
}

extern "C" int vis_gemm_decode_fp8_ksplit(int N, int K) {
  if (N <= 0 || K < 128) return 0;
  int spb, slots, nwg;
  gemm3_geometry(N, K, false, &spb, &slots, &nwg, 128);
  return slots;
}

// Split-K second half: C = act(sum_slices part + bias) + R, bf16 out; 8 columns per thread, coalesced.
__global__ __launch_bounds__(256) void gemm_splitk_finalize_kernel(GemmArgs p) {
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
    if (p.bias) {
      float f[8];
      unpack8(*(const u32x4*)(p.bias + n), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += f[e];
    }
    if (p.act != ACT_NONE) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = act_apply(v[e], p.act);
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

// Tile choice, measured on MI355X (tools/gemm_bench.py, tools/kbench.py).  VIS_GEMM_TILE=1|2|4 forces a shape.
//  * 256x256 (1 WG/CU): best per-tile rate; used when its last round of 256 CUs is at least half full, and for
//    wide problems (>= 3 rounds) with the ragged remainder columns handed to a second launch (LLM gate/up:
//    9 x 148 tiles = 5.2 rounds -> 142 columns here + 6 columns below, instead of 6 rounds);
//  * 256x192 / 256x128 forms of the same pipelined kernel (NT = 3 / 2): problems whose grid is then one (nearly) full
//    round - LLM qkv 9 x 24 = 216 tiles (0.095 -> 0.071 ms against 1.27 rounds of 128x128 tiles), LLM o 9 x 28 = 252,
//    ViT fc2 20 x 10 = 200; the older 3-stage 256x128 kernel (VIS_GEMM_TILE=2) is 3-5 % slower and kept for A/B;
//  * 128x256 ping-pong half-tiles (1 WG/CU): what would otherwise go to 128x128 tiles, when the half-tile grid needs
//    no more rounds (LLM o, ViT proj, the gate/up remainder columns, the merger);
//  * 128x128 (2 WG/CU): everything else.
static int gemm_dispatch(GemmArgs p, hipStream_t stream) {
  static const int forced = [] { const char* e = getenv("VIS_GEMM_TILE"); return e ? atoi(e) : 0; }();
  const int M = p.M, N = p.N, K = p.K;
  const int tn1 = (N + GEMM_BN - 1) / GEMM_BN;
  const int t2 = ((M + GEMM2_BM - 1) / GEMM2_BM) * tn1;
  const int tm4 = (M + GEMM4_B - 1) / GEMM4_B, tn4 = (N + GEMM4_B - 1) / GEMM4_B;
  const int t4 = tm4 * tn4, last = t4 % 256;
  const bool big = forced == 2;   // the 3-stage 256x128 kernel: A/B only
  static const int pp_env = [] { const char* e = getenv("VIS_GEMM_PP"); return e ? atoi(e) : 1; }();   // 0: the 2-phase kernel (A/B)
  const bool use_pp = (pp_env != 0 && forced != 4) || forced == 7;
  // Kernel choice by a two-line cost model fitted to cold-cache timings of the production shapes (tools/gemm_tiles_ab.py,
  // r02): a 256x256 ping-pong tile costs ~1.41 us per 64-wide K-step + 4.3 us (prologue + epilogue), one per CU; a
  // 128x128 tile ~1.2 us per K-step + 3 us with two per CU.  Rounds are what ragged grids pay for:
  //   ViT qkv  4900x3840x1280: 300 tiles of 256^2 = 2 rounds (66 us) beat 1170 of 128^2 = 3 rounds (78 us);
  //   LLM qkv  2249x4608x3584: 162 tiles of 256^2 on 63 % of the CUs (83 us) still beat 216 tiles of 256x192 (90 us);
  //   LLM o    2249x3584x3584: 504 tiles of 128^2 = one full round of two per CU (75 us) beat 252 of 256x128 (85 us).
  // Every kernel accumulates K in the same order, so the choice (which depends on M) never changes a result bit.
  const int nk64 = K / GEMM_BK;
  const int t1 = ((M + GEMM_BM - 1) / GEMM_BM) * tn1;
  const float cost1 = (float)((t1 + 511) / 512) * (1.2f * nk64 + 3.0f);
  const float cost4 = (float)((t4 + 255) / 256) * (1.41f * nk64 + 4.3f);
  bool huge = forced ? (forced == 4 || forced == 7) : (K >= 256 && M > 128 && N > 128 && cost4 < cost1);
  int cols4 = tn4;  // 256-wide tile columns given to the 256x256 kernel
  // ... and (r05) ONE whole round + a remainder of fewer than 64 tiles: the second round of a 294-tile grid (LLM o at four images)
  // or a 264-tile one (the Auditor's qkv) costs almost a full round for 15 % / 3 % of the tiles - 155 -> 141 us and 162 -> 141 us
  // with the remainder columns on half-tiles (tools/probes/gemm_mix_probe.py, launches alone; inside `dual` at 32 per step the
  // difference is within run-to-run noise, 9.93 vs 9.93 images/s; VIS_GEMM_MIX1=0: off, A/B)
  static const int mix1 = [] { const char* e = getenv("VIS_GEMM_MIX1"); return e ? atoi(e) : 64; }();
  if (!forced && K >= 1024 && M >= 1024 && last != 0 && ((t4 >= 512 && last < 128) || (t4 > 256 && t4 < 512 && last < mix1))) {   // whole rounds + a thin remainder
    cols4 = (t4 / 256) * 256 / tm4;  // whole rounds only; the ragged remainder columns go through this function again
    huge = cols4 > 0;
  }
  // 256 x 192 / 256 x 128 forms of the 2-phase pipelined kernel: A/B only (r02: never the fastest on a production shape)
  const bool nt3 = forced == 6;
  const bool nt2 = forced == 5;
  if (nt3 || nt2) {
    static const bool attr_ok = [] {
      return hipFuncSetAttribute((const void*)gemm_bf16_256xN_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 GEMM4_LDS_BYTES) == hipSuccess &&
             hipFuncSetAttribute((const void*)gemm_bf16_256xN_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 GEMM4_LDS_BYTES) == hipSuccess;
    }();
    if (!attr_ok) return VIS_ERR_LAUNCH;
    p.tiles_m = tm4;
    if (nt3) {
      p.tiles_n = (N + 191) / 192;
      hipLaunchKernelGGL(gemm_bf16_256xN_kernel<3>, dim3(p.tiles_m * p.tiles_n), dim3(512), GEMM4_LDS_BYTES, stream, p);
    } else {
      p.tiles_n = tn1;
      hipLaunchKernelGGL(gemm_bf16_256xN_kernel<2>, dim3(p.tiles_m * p.tiles_n), dim3(512), GEMM4_LDS_BYTES, stream, p);
    }
    return VIS_OK;
  }
  // 128 x 256 ping-pong half-tiles wherever the 128 x 128 kernel would run and the half-tile grid needs no more rounds
  // than its grid does (LLM o: 252 half-tiles = one round, 74.9 -> 68.0 us; ViT proj 195: 36.4 -> 33.6; the gate/up
  // remainder 108: 62.0 -> 56.6 - tools/gemm_tiles_ab.py, cold caches).  Its K-tile costs 1.14 us against the 128 x 128
  // tile's 1.2 us for HALF the work: both sit at the CU's LDS bandwidth (176 KiB of fragment reads + LDS-DMA writes per
  // K-tile here), which is why the gain is 9 % and not the 40 % the MFMA count alone would give.  VIS_GEMM_HALF=0: off.
  static const bool half_ok = [] { const char* e = getenv("VIS_GEMM_HALF"); return !(e && atoi(e) == 0); }();
  const int t6 = ((M + 127) / 128) * tn4;
  const bool half = forced == 8 || (!forced && half_ok && !huge && M > 128 && N >= 256 && K >= 512 &&
                                    (t6 + 255) / 256 <= (t1 + 511) / 512);
  if (half) {
    static const bool attr6_ok = [] {
      return hipFuncSetAttribute((const void*)gemm_bf16_128x256_pp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 GEMM6_LDS_BYTES) == hipSuccess;
    }();
    if (!attr6_ok) return VIS_ERR_LAUNCH;
    p.tiles_m = (M + 127) / 128;
    p.tiles_n = tn4;
    hipLaunchKernelGGL(gemm_bf16_128x256_pp_kernel, dim3(p.tiles_m * p.tiles_n), dim3(512), GEMM6_LDS_BYTES, stream, p);
    return VIS_OK;
  }
  if (huge) {
    static const bool attr4_ok = [] {
      return hipFuncSetAttribute((const void*)gemm_bf16_256xN_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 GEMM4_LDS_BYTES) == hipSuccess;
    }();
    if (!attr4_ok) return VIS_ERR_LAUNCH;
    GemmArgs q = p;
    if (cols4 < tn4) q.N = cols4 * GEMM4_B;
    q.tiles_m = tm4;
    q.tiles_n = cols4;
    if (use_pp) {
      static const bool attr5_ok = [] {
        return hipFuncSetAttribute((const void*)gemm_bf16_256x256_pp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   GEMM5_LDS_BYTES) == hipSuccess;
      }();
      if (!attr5_ok) return VIS_ERR_LAUNCH;
      hipLaunchKernelGGL(gemm_bf16_256x256_pp_kernel, dim3(q.tiles_m * q.tiles_n), dim3(512), GEMM5_LDS_BYTES, stream, q);
    } else {
      hipLaunchKernelGGL(gemm_bf16_256xN_kernel<4>, dim3(q.tiles_m * q.tiles_n), dim3(512), GEMM4_LDS_BYTES, stream, q);
    }
    if (cols4 == tn4) return VIS_OK;
    // remainder columns [n_off, N): same problem, shifted operands
    const int n_off = cols4 * GEMM4_B;
    const int c_off = (p.act == ACT_SWIGLU) ? n_off / 2 : n_off;
    p.W += (size_t)n_off * p.ldw;
    if (p.bias) p.bias += n_off;
    if (p.R) p.R += c_off;
    p.C += c_off;
    p.N = N - n_off;
    return gemm_dispatch(p, stream);
  }
  p.tiles_n = tn1;
  if (big) {
    static const bool attr_ok = [] {
      return hipFuncSetAttribute((const void*)gemm_bf16_256x128_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 GEMM2_LDS_BYTES) == hipSuccess;
    }();
    if (!attr_ok) return VIS_ERR_LAUNCH;
    p.tiles_m = (M + GEMM2_BM - 1) / GEMM2_BM;
    hipLaunchKernelGGL(gemm_bf16_256x128_kernel, dim3(p.tiles_m * p.tiles_n), dim3(512), GEMM2_LDS_BYTES, stream, p);
  } else {
    p.tiles_m = (M + GEMM_BM - 1) / GEMM_BM;
    hipLaunchKernelGGL(gemm_bf16_128x128_kernel, dim3(p.tiles_m * p.tiles_n), dim3(256), 0, stream, p);
  }
  return VIS_OK;
}

// C-ABI launcher (declared in include/vis_hip.h)
extern "C" int vis_gemm_bf16(const void* A, const void* W, const void* bias, const void* R, void* C,
                             int M, int N, int K, int lda, int ldw, int ldc, int ldr, int act,
                             hipStream_t stream) {
  if (!A || !W || !C || M <= 0 || N <= 0 || K <= 0) return VIS_ERR_ARG;
  if (K % GEMM_BK != 0 || N % 4 != 0) return VIS_ERR_ARG;
  if (lda % 8 != 0 || ldw % 8 != 0 || ldc % 4 != 0 || (R && ldr % 4 != 0)) return VIS_ERR_ARG;
  if (act < ACT_NONE || act > ACT_SWIGLU) return VIS_ERR_ARG;
  if (act == ACT_SWIGLU && (N % 32 != 0 || R)) return VIS_ERR_ARG;   // bias (if any) is interleaved like the weight rows
  if (((uintptr_t)A | (uintptr_t)W | (uintptr_t)C | (uintptr_t)bias | (uintptr_t)R) & 7) return VIS_ERR_ARG;
  if (((uintptr_t)A | (uintptr_t)W) & 15) return VIS_ERR_ARG;
  GemmArgs p;
  p.A = (const bf16_t*)A;
  p.W = (const bf16_t*)W;
  p.bias = (const bf16_t*)bias;
  p.R = (const bf16_t*)R;
  p.C = (bf16_t*)C;
  p.M = M; p.N = N; p.K = K;
  p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.ldr = ldr;
  p.act = act;
  p.part = nullptr; p.ksplit = 1; p.out_f32 = 0;
  static const int wide_env = [] { const char* e = getenv("VIS_GEMM_WIDE"); return e ? atoi(e) : 1; }();   // 0: direct epilogue (A/B)
  p.wide = wide_env && N % 8 == 0 && ldc % 8 == 0 && (!R || ldr % 8 == 0) && !(((uintptr_t)C | (uintptr_t)R) & 15) &&
           !(act == ACT_SWIGLU && N % 16 != 0);
  static const int nt_env = [] { const char* e = getenv("VIS_GEMM_NT"); return e ? atoi(e) : 1; }();   // 0 never, 1 by size, 2 always (A/B)
  p.nt = nt_env == 2 || (nt_env == 1 && (size_t)M * (act == ACT_SWIGLU ? N / 2 : N) * 2 >= ((size_t)64 << 20));
  vis_clear_error();
  const int st = gemm_dispatch(p, stream);
  return st != VIS_OK ? st : vis_check_launch();
}

// Split-K form of vis_gemm_bf16 for long-K problems whose 256x256 tile count is too small to fill the chip
// (LLM down projection: 9 x 14 = 126 tiles, K = 18944): `ksplit` K-slices run as independent 256x256 tiles
// (126 x 2 = 252 workgroups), f32 partials go to `work` (ksplit x M x N floats), a second launch sums them and
// applies bias / activation / residual.  SwiGLU is not supported here.
static int gemm_splitk_launch(const void* A, const void* W, const void* bias, const void* R, void* C, void* work,
                              int M, int N, int K, int lda, int ldw, int ldc, int ldr, int act, int ksplit,
                              bool finalize, hipStream_t stream) {
  if (!A || !W || !work || M <= 0 || N <= 0 || K <= 0) return VIS_ERR_ARG;
  if (finalize && !C) return VIS_ERR_ARG;
  if (K % GEMM_BK != 0 || N % 8 != 0) return VIS_ERR_ARG;
  if (lda % 8 != 0 || ldw % 8 != 0 || (finalize && (ldc % 8 != 0 || (R && ldr % 8 != 0)))) return VIS_ERR_ARG;
  if (act < ACT_NONE || act >= ACT_SWIGLU) return VIS_ERR_ARG;
  if (ksplit < 2 || ksplit > 8 || K / GEMM_BK < 2 * ksplit) return VIS_ERR_ARG;
  if (((uintptr_t)A | (uintptr_t)W | (uintptr_t)C | (uintptr_t)bias | (uintptr_t)R | (uintptr_t)work) & 15) return VIS_ERR_ARG;
  static const bool attr4_ok = [] {
    return hipFuncSetAttribute((const void*)gemm_bf16_256xN_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               GEMM4_LDS_BYTES) == hipSuccess;
  }();
  if (!attr4_ok) return VIS_ERR_LAUNCH;
  GemmArgs p;
  p.A = (const bf16_t*)A; p.W = (const bf16_t*)W; p.bias = (const bf16_t*)bias; p.R = (const bf16_t*)R; p.C = (bf16_t*)C;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.ldr = ldr; p.act = act;
  p.part = (float*)work; p.ksplit = ksplit; p.out_f32 = 0; p.wide = 0; p.nt = 0;
  p.tiles_m = (M + GEMM4_B - 1) / GEMM4_B;
  p.tiles_n = (N + GEMM4_B - 1) / GEMM4_B;
  vis_clear_error();
  static const int pp_env = [] { const char* e = getenv("VIS_GEMM_PP"); return e ? atoi(e) : 1; }();
  if (pp_env) {
    static const bool attr5_ok = [] {
      return hipFuncSetAttribute((const void*)gemm_bf16_256x256_pp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 GEMM5_LDS_BYTES) == hipSuccess;
    }();
    if (!attr5_ok) return VIS_ERR_LAUNCH;
    hipLaunchKernelGGL(gemm_bf16_256x256_pp_kernel, dim3(p.tiles_m * p.tiles_n, ksplit), dim3(512), GEMM5_LDS_BYTES, stream, p);
  } else {
    hipLaunchKernelGGL(gemm_bf16_256xN_kernel<4>, dim3(p.tiles_m * p.tiles_n, ksplit), dim3(512), GEMM4_LDS_BYTES, stream, p);
  }
  if (finalize) {
    const long long total = (long long)M * (N / 8);
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(gemm_splitk_finalize_kernel, dim3(blocks), dim3(256), 0, stream, p);
  }
  return vis_check_launch();
}

extern "C" int vis_gemm_bf16_splitk(const void* A, const void* W, const void* bias, const void* R, void* C, void* work,
                                    int M, int N, int K, int lda, int ldw, int ldc, int ldr, int act, int ksplit,
                                    hipStream_t stream) {
  return gemm_splitk_launch(A, W, bias, R, C, work, M, N, K, lda, ldw, ldc, ldr, act, ksplit, true, stream);
}

// The K-sliced tiles alone: work[ksplit][M][N] f32 partial sums of A * W^T; the caller finalises them
// (vis_splitk_finalize_norm: sum + bias + residual fused with the next RMSNorm / LayerNorm).
extern "C" int vis_gemm_bf16_splitk_part(const void* A, const void* W, void* work, int M, int N, int K, int lda, int ldw,
                                         int ksplit, hipStream_t stream) {
  return gemm_splitk_launch(A, W, nullptr, nullptr, nullptr, work, M, N, K, lda, ldw, 0, 0, ACT_NONE, ksplit, false, stream);
}
