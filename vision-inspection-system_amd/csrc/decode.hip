// K10 / K11 / K12: the per-token decode path on gfx950 - HBM-bound by design.
//
//  * vis_gemv_bf16: y = act(W x + bias) + R with an optional fused RMSNorm of x
//    (K3 folded into the prologue).  Weights are streamed exactly once,
//    16 B per lane straight to VGPRs (no LDS round trip for a read-once
//    operand), x lives in LDS, products via v_dot2c_f32_bf16, f32 accumulate.
//    Algorithmic bytes per launch: N*K*2 (weights) - the roofline denominator
//    of the whole decode phase (14.14 GB per generated token at 7B).
//  * vis_decode_attn (+ combine): GQA attention of one query token over the KV
//    cache, split over the context so the 4 KV heads still fill the chip.
//  * vis_argmax_f32: greedy next-token pick, appends to the device-side token
//    list and advances the device-side step counter, so a whole decode step
//    needs no host round trip and sits in one replayable hipGraph.
// Reference semantics: TF:models/qwen2_vl/modeling_qwen2_vl.py:96-110 (norm),
// :459-466 (MLP), :501-556 (attention), greedy = argmax over lm_head logits.
#include "decode_common.hip.h"
#include <stdlib.h>

// NB = input rows sharing the weight stream (1: the single-sequence decode step; 2 / 4: vis_gemv_bf16_rows, a handful of
// in-flight sequences - each row's arithmetic is exactly the NB = 1 kernel's, so its result is bit-identical to it)
// AMAX (lm_head of the single-sequence step): the greedy / Gumbel-max pick's first stage rides in the epilogue - every
// workgroup leaves the (value, first index) maximum of ITS rows in amax_val / amax_idx[blockIdx.x] (argmax_stage1_kernel's
// comparison, so the pick is the one the two-stage form makes), saving a launch per token.
template <int NB, bool AMAX = false>
__global__ __launch_bounds__(256) void gemv_bf16_kernel(GemvArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* xs = (bf16_t*)smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nch = p.K >> 3;

  const bool swiglu = (p.act == GV_ACT_SWIGLU);
  const int n_pairs = swiglu ? (p.N >> 1) : ((p.N + 1) >> 1);
  const int n_waves = gridDim.x * 4;
  const int wid = blockIdx.x * 4 + wave;
  const int p_begin = (int)((long long)n_pairs * wid / n_waves);
  const int p_end = (int)((long long)n_pairs * (wid + 1) / n_waves);
  const int nseg = ((nch + 63) / 64 + GV_SEG - 1) / GV_SEG;
  const int n_tasks = (p_end - p_begin) * nseg;

  GvBuf A, B;
  if (n_tasks > 0) gv_load(A, p, swiglu, p_begin, 0, lane, nch);  // in flight while x is staged

  // ---- stage x (optionally RMS-normalised) into LDS; rows past nb repeat the last one (computed, never stored)
#pragma unroll
  for (int r = 0; r < NB; ++r) {
    if (r > 0) __syncthreads();   // the norm's reduction scratch is reused
    gv_stage_row(p.x + (size_t)min(r, p.nb - 1) * p.ldx, p.norm_w, xs + (size_t)r * p.K, nch, p.K, p.eps, tid, lane, wave);
  }
  __syncthreads();

  float a0[NB], a1[NB];
#pragma unroll
  for (int r = 0; r < NB; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
  float best = -INFINITY;       // AMAX: lane 0's running maximum over this wave's rows
  int bi = 0x7fffffff;
  const unsigned am_step = AMAX ? (unsigned)*p.am_step : 0u;
  auto amax_rows = [&](int pr) {      // after gv_finish: a0[0] / a1[0] hold the two rows' sums (= the stored logits)
    if (lane != 0) return;
    const int o = 2 * pr;
    float v0 = a0[0], v1 = a1[0];
    if (p.am_inv_temp > 0.f) {
      v0 = v0 * p.am_inv_temp + gumbel_noise(p.am_seed, am_step, (unsigned)o);
      v1 = v1 * p.am_inv_temp + gumbel_noise(p.am_seed, am_step, (unsigned)(o + 1));
    }
    if (v0 > best || (v0 == best && o < bi)) { best = v0; bi = o; }
    if (o + 1 < p.N && (v1 > best || (v1 == best && o + 1 < bi))) { best = v1; bi = o + 1; }
  };
  int pair = p_begin, seg = 0;  // task being consumed
  for (int t = 0; t < n_tasks; t += 2) {
    // next task (t+1) -> B
    int pair1 = pair, seg1 = seg + 1;
    if (seg1 == nseg) { seg1 = 0; ++pair1; }
    if (t + 1 < n_tasks) gv_load(B, p, swiglu, pair1, seg1, lane, nch);
    gv_consume<NB>(A, xs, p.K, seg, lane, nch, a0, a1);
    if (seg == nseg - 1) {
      gv_finish<NB>(p, swiglu, pair, lane, a0, a1);
      if constexpr (AMAX) amax_rows(pair);
#pragma unroll
      for (int r = 0; r < NB; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
    }
    if (t + 1 >= n_tasks) break;
    // task t+2 -> A
    int pair2 = pair1, seg2 = seg1 + 1;
    if (seg2 == nseg) { seg2 = 0; ++pair2; }
    if (t + 2 < n_tasks) gv_load(A, p, swiglu, pair2, seg2, lane, nch);
    gv_consume<NB>(B, xs, p.K, seg1, lane, nch, a0, a1);
    if (seg1 == nseg - 1) {
      gv_finish<NB>(p, swiglu, pair1, lane, a0, a1);
      if constexpr (AMAX) amax_rows(pair1);
#pragma unroll
      for (int r = 0; r < NB; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
    }
    pair = pair2;
    seg = seg2;
  }
  if constexpr (AMAX) {
    __shared__ float am_v[4];
    __shared__ int am_i[4];
    if (lane == 0) { am_v[wave] = best; am_i[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < 4; ++w)
        if (am_v[w] > best || (am_v[w] == best && am_i[w] < bi)) { best = am_v[w]; bi = am_i[w]; }
      p.am_val[blockIdx.x] = best;
      p.am_idx[blockIdx.x] = bi;
    }
  }
}

// LDS of a multi-row launch: nb rows of K bf16 (the attribute is raised once per kernel instance)
#define GV_ROWS_LDS_MAX (152 * 1024)

static int gemv_bf16_launch(GemvArgs p, hipStream_t stream) {
  const int n_pairs = (p.act == GV_ACT_SWIGLU) ? p.N / 2 : (p.N + 1) / 2;
  // one row pair per wave until the grid reaches ~4096 waves, then several pairs per wave
  int blocks = (n_pairs + 3) / 4;
  if (blocks > 1024) blocks = 1024 + (blocks - 1024) / 8;
  if (blocks > 2048) blocks = 2048;
  vis_clear_error();
  if (p.nb == 1) {
    hipLaunchKernelGGL(gemv_bf16_kernel<1>, dim3(blocks), dim3(256), (size_t)p.K * 2, stream, p);
  } else {
    static const bool attr_ok = [] {
      return hipFuncSetAttribute((const void*)gemv_bf16_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, GV_ROWS_LDS_MAX) == hipSuccess &&
             hipFuncSetAttribute((const void*)gemv_bf16_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, GV_ROWS_LDS_MAX) == hipSuccess;
    }();
    if (!attr_ok) return VIS_ERR_LAUNCH;
    if (p.nb == 2) hipLaunchKernelGGL(gemv_bf16_kernel<2>, dim3(blocks), dim3(256), (size_t)p.K * 4, stream, p);
    else hipLaunchKernelGGL(gemv_bf16_kernel<4>, dim3(blocks), dim3(256), (size_t)p.K * 8, stream, p);
  }
  return vis_check_launch();
}

extern "C" int vis_gemv_bf16(const void* x, const void* W, const void* bias, const void* R, const void* norm_w,
                             void* y, int N, int K, int ldw, int act, int out_f32, float eps,
                             hipStream_t stream) {
  if (!x || !W || !y || N <= 0 || K <= 0) return VIS_ERR_ARG;
  if (K % 8 != 0 || ldw % 8 != 0 || K * 2 > 60 * 1024) return VIS_ERR_ARG;
  if (act != GV_ACT_NONE && act != GV_ACT_SWIGLU) return VIS_ERR_ARG;
  if (act == GV_ACT_SWIGLU && (N % 32 != 0 || bias || R || out_f32)) return VIS_ERR_ARG;
  if (((uintptr_t)x | (uintptr_t)W | (uintptr_t)norm_w) & 15) return VIS_ERR_ARG;
  GemvArgs p;
  p.x = (const bf16_t*)x; p.W = (const bf16_t*)W; p.bias = (const bf16_t*)bias; p.R = (const bf16_t*)R;
  p.norm_w = (const bf16_t*)norm_w; p.y = y;
  p.N = N; p.K = K; p.ldw = ldw; p.act = act; p.out_f32 = out_f32; p.eps = eps;
  p.outs_per_block = 0;
  p.nb = 1; p.ldx = 0; p.ldy = 0; p.ldr = 0;
  return gemv_bf16_launch(p, stream);
}

// K10 for a handful of in-flight sequences (B <= 4): y[b] = act(W x[b] + bias) + R[b], optional fused RMSNorm of every
// x row - the weights are streamed ONCE for all rows, each row's arithmetic is vis_gemv_bf16's (bit-identical results),
// and there is no partial buffer and no finalisation launch (vis_gemm_decode_bf16 + vis_skinny_finalize need both).
extern "C" int vis_gemv_bf16_rows(const void* x, const void* W, const void* bias, const void* R, const void* norm_w,
                                  void* y, int B, int N, int K, int ldw, int ldx, int ldy, int ldr, int act, int out_f32,
                                  float eps, hipStream_t stream) {
  if (!x || !W || !y || N <= 0 || K <= 0 || B < 1 || B > 4) return VIS_ERR_ARG;
  if (K % 8 != 0 || ldw % 8 != 0 || ldx % 8 != 0 || ldx < K) return VIS_ERR_ARG;
  const int nbk = (B == 1) ? 1 : (B == 2 ? 2 : 4);
  if ((size_t)K * 2 * nbk > (size_t)(nbk == 1 ? 60 * 1024 : GV_ROWS_LDS_MAX)) return VIS_ERR_ARG;
  if (act != GV_ACT_NONE && act != GV_ACT_SWIGLU) return VIS_ERR_ARG;
  if (act == GV_ACT_SWIGLU && (N % 32 != 0 || bias || R || out_f32)) return VIS_ERR_ARG;
  if (ldy < ((act == GV_ACT_SWIGLU) ? N / 2 : N) || (R && ldr < N)) return VIS_ERR_ARG;
  if (((uintptr_t)x | (uintptr_t)W | (uintptr_t)norm_w) & 15) return VIS_ERR_ARG;
  GemvArgs p;
  p.x = (const bf16_t*)x; p.W = (const bf16_t*)W; p.bias = (const bf16_t*)bias; p.R = (const bf16_t*)R;
  p.norm_w = (const bf16_t*)norm_w; p.y = y;
  p.N = N; p.K = K; p.ldw = ldw; p.act = act; p.out_f32 = out_f32; p.eps = eps;
  p.outs_per_block = 0;
  p.nb = B; p.ldx = ldx; p.ldy = ldy; p.ldr = ldr;
  return gemv_bf16_launch(p, stream);   // three rows run on the four-row kernel (the fourth repeats the third, not stored)
}

// ---------------------------------------------------------------------------
// vis_gemv_fp8w (BASELINE configs[4] slice: fp8 weights for the HBM-bound decode GEMVs, "W8A16"):
//   y = act((Wq x) * scale + bias) + R,   Wq = OCP e4m3 bytes [N][ldw], scale f32 [N] (per output row)
// Same structure as vis_gemv_bf16 (x in LDS as bf16, optional fused RMSNorm prologue, weights streamed exactly
// once with non-temporal 16-byte loads, two register sets in flight), but a 16-byte load now carries 16 weights:
// v_cvt_scalef32_pk_bf16_fp8 turns each fp8 pair into a bf16 pair EXACTLY (e4m3 has 3 mantissa bits), which feeds
// the same v_dot2c_f32_bf16.  A task is FOUR rows (two row pairs / two SwiGLU outputs) x 4 chunks per lane, so the
// bytes in flight per lane match the bf16 kernel although a row is half as long.
// Algorithmic bytes per launch: N*K (weights) + 4 N (scales).
// Task shape <ROWS, SEG>: ROWS weight rows x SEG 16-byte chunks per lane per register set.  Short rows (K = 3584:
// 3.5 chunks per lane) use <4, 4>; projections with few, long rows (down: N = 3584, K = 18944) use <2, 8> so that
// the grid still has ~7 waves per CU.

struct GemvF8Args {
  const bf16_t* x;
  const uint8_t* W;
  const float* scale;
  const bf16_t* bias;
  const bf16_t* R;
  const bf16_t* norm_w;
  void* y;
  int N, K, ldw;
  int act, out_f32;
  float eps;
  int nb, ldx, ldy, ldr;   // vis_gemv_fp8w_rows: as in GemvArgs
};

template <int ROWS, int SEG>
struct GfBuf {
  u32x4 w[ROWS][SEG];
};

template <int ROWS>
__device__ __forceinline__ void gf_rows(const GemvF8Args& p, bool swiglu, int quad, int* r) {
  if (swiglu) {  // ROWS/2 consecutive outputs of one 16-group: gate rows r[0], r[2], ..., up rows r[1], r[3], ...
    const int o = (ROWS / 2) * quad;
    const int g0 = ((o >> 4) << 5) + (o & 15);
#pragma unroll
    for (int i = 0; i < ROWS / 2; ++i) { r[2 * i] = g0 + i; r[2 * i + 1] = g0 + 16 + i; }
  } else {
#pragma unroll
    for (int i = 0; i < ROWS; ++i) r[i] = min(ROWS * quad + i, p.N - 1);
  }
}

template <int ROWS, int SEG>
__device__ __forceinline__ void gf_load(GfBuf<ROWS, SEG>& b, const GemvF8Args& p, bool swiglu, int quad, int seg,
                                        int lane, int nch) {
  int r[ROWS];
  gf_rows<ROWS>(p, swiglu, quad, r);
#pragma unroll
  for (int i = 0; i < ROWS; ++i) {
    const uint8_t* w = p.W + (size_t)r[i] * p.ldw;
#pragma unroll
    for (int u = 0; u < SEG; ++u) {
      const int c = min(lane + 64 * (seg * SEG + u), nch - 1);  // unconditional, clamped (see gv_load)
      b.w[i][u] = __builtin_nontemporal_load((const u32x4*)(w + c * 16));
    }
  }
}

// the 16 weights of a 16-byte load as eight EXACT bf16 pairs (converted once, used for every input row)
struct Gf16 {
  bf16x2 v[8];
};
__device__ __forceinline__ Gf16 cvt16_f8(const u32x4& w) {
  Gf16 r;
  // word i of w holds weights 4i..4i+3: low half -> k = 4i, 4i+1; high half -> k = 4i+2, 4i+3
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    r.v[2 * i] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w[i], 1.0f, false);
    r.v[2 * i + 1] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w[i], 1.0f, true);
  }
  return r;
}
__device__ __forceinline__ float dot16_f8(const Gf16& w, const u32x4& xlo, const u32x4& xhi, float acc) {
  const bf16x8 xa = __builtin_bit_cast(bf16x8, xlo), xb = __builtin_bit_cast(bf16x8, xhi);
  acc = __builtin_amdgcn_fdot2_f32_bf16(w.v[0], __builtin_shufflevector(xa, xa, 0, 1), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(w.v[1], __builtin_shufflevector(xa, xa, 2, 3), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(w.v[2], __builtin_shufflevector(xa, xa, 4, 5), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(w.v[3], __builtin_shufflevector(xa, xa, 6, 7), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(w.v[4], __builtin_shufflevector(xb, xb, 0, 1), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(w.v[5], __builtin_shufflevector(xb, xb, 2, 3), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(w.v[6], __builtin_shufflevector(xb, xb, 4, 5), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(w.v[7], __builtin_shufflevector(xb, xb, 6, 7), acc, false);
  return acc;
}

template <int ROWS, int SEG, int NB>
__device__ __forceinline__ void gf_consume(const GfBuf<ROWS, SEG>& b, const bf16_t* xs, int K, int seg, int lane, int nch,
                                           float (&a)[NB][ROWS]) {
#pragma unroll
  for (int u = 0; u < SEG; ++u) {
    const int c = lane + 64 * (seg * SEG + u);
    const int cc = min(c, nch - 1);
    u32x4 xlo[NB], xhi[NB];
#pragma unroll
    for (int r = 0; r < NB; ++r) {
      xlo[r] = *(const u32x4*)(xs + (size_t)r * K + cc * 16);
      xhi[r] = *(const u32x4*)(xs + (size_t)r * K + cc * 16 + 8);
      if (c >= nch) { xlo[r] = (u32x4){0u, 0u, 0u, 0u}; xhi[r] = xlo[r]; }  // clamped duplicate chunk contributes nothing
    }
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
      const Gf16 wv = cvt16_f8(b.w[i][u]);
#pragma unroll
      for (int r = 0; r < NB; ++r) a[r][i] = dot16_f8(wv, xlo[r], xhi[r], a[r][i]);
    }
  }
}

template <int ROWS, int NB>
__device__ __forceinline__ void gf_finish(const GemvF8Args& p, bool swiglu, int quad, int lane, float (&a)[NB][ROWS]) {
#pragma unroll
  for (int r = 0; r < NB; ++r)
#pragma unroll
    for (int i = 0; i < ROWS; ++i) a[r][i] = wave_sum(a[r][i]);
  if (lane != 0) return;
  int rw[ROWS];
  gf_rows<ROWS>(p, swiglu, quad, rw);
#pragma unroll
  for (int r = 0; r < NB; ++r) {
    if (r >= p.nb) break;
    if (swiglu) {
#pragma unroll
      for (int i = 0; i < ROWS / 2; ++i) {
        const float g = a[r][2 * i] * p.scale[rw[2 * i]], u = a[r][2 * i + 1] * p.scale[rw[2 * i + 1]];
        ((bf16_t*)p.y)[(size_t)r * p.ldy + (ROWS / 2) * quad + i] = f2bf(silu_fast(g) * u);
      }
      continue;
    }
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
      const int o = ROWS * quad + i;
      if (o >= p.N) break;
      float v = a[r][i] * p.scale[o];
      if (p.bias) v += bf2f(p.bias[o]);
      if (p.R) v += bf2f(p.R[(size_t)r * p.ldr + o]);
      if (p.out_f32) ((float*)p.y)[(size_t)r * p.ldy + o] = v;
      else ((bf16_t*)p.y)[(size_t)r * p.ldy + o] = f2bf(v);
    }
  }
}

template <int ROWS, int SEG, int NB>
__global__ __launch_bounds__(256) void gemv_fp8w_kernel(GemvF8Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* xs = (bf16_t*)smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nch8 = p.K >> 3;   // 16-byte chunks of x (bf16)
  const int nch = p.K >> 4;    // 16-byte chunks of a weight row (fp8)
  const bool swiglu = (p.act == GV_ACT_SWIGLU);
  const int n_out = swiglu ? (p.N >> 1) : p.N;
  const int n_quads = swiglu ? (n_out / (ROWS / 2)) : ((n_out + ROWS - 1) / ROWS);
  const int n_waves = gridDim.x * 4;
  const int wid = blockIdx.x * 4 + wave;
  const int q_begin = (int)((long long)n_quads * wid / n_waves);
  const int q_end = (int)((long long)n_quads * (wid + 1) / n_waves);
  const int nseg = ((nch + 63) / 64 + SEG - 1) / SEG;
  const int n_tasks = (q_end - q_begin) * nseg;

  GfBuf<ROWS, SEG> A, B;
  if (n_tasks > 0) gf_load(A, p, swiglu, q_begin, 0, lane, nch);  // in flight while x is staged

#pragma unroll
  for (int r = 0; r < NB; ++r) {
    if (r > 0) __syncthreads();   // the norm's reduction scratch is reused
    gv_stage_row(p.x + (size_t)min(r, p.nb - 1) * p.ldx, p.norm_w, xs + (size_t)r * p.K, nch8, p.K, p.eps, tid, lane, wave);
  }
  __syncthreads();

  float acc[NB][ROWS];
#pragma unroll
  for (int r = 0; r < NB; ++r)
#pragma unroll
    for (int i = 0; i < ROWS; ++i) acc[r][i] = 0.f;
  int quad = q_begin, seg = 0;
  for (int t = 0; t < n_tasks; t += 2) {
    int quad1 = quad, seg1 = seg + 1;
    if (seg1 == nseg) { seg1 = 0; ++quad1; }
    if (t + 1 < n_tasks) gf_load(B, p, swiglu, quad1, seg1, lane, nch);
    gf_consume<ROWS, SEG, NB>(A, xs, p.K, seg, lane, nch, acc);
    if (seg == nseg - 1) {
      gf_finish<ROWS, NB>(p, swiglu, quad, lane, acc);
#pragma unroll
      for (int r = 0; r < NB; ++r)
#pragma unroll
        for (int i = 0; i < ROWS; ++i) acc[r][i] = 0.f;
    }
    if (t + 1 >= n_tasks) break;
    int quad2 = quad1, seg2 = seg1 + 1;
    if (seg2 == nseg) { seg2 = 0; ++quad2; }
    if (t + 2 < n_tasks) gf_load(A, p, swiglu, quad2, seg2, lane, nch);
    gf_consume<ROWS, SEG, NB>(B, xs, p.K, seg1, lane, nch, acc);
    if (seg1 == nseg - 1) {
      gf_finish<ROWS, NB>(p, swiglu, quad1, lane, acc);
#pragma unroll
      for (int r = 0; r < NB; ++r)
#pragma unroll
        for (int i = 0; i < ROWS; ++i) acc[r][i] = 0.f;
    }
    quad = quad2;
    seg = seg2;
  }
}

static int gemv_fp8w_launch(GemvF8Args p, hipStream_t stream) {
  const int n_out = (p.act == GV_ACT_SWIGLU) ? p.N / 2 : p.N;
  static const int forced = [] { const char* e = getenv("VIS_GEMV8_SHAPE"); return e ? atoi(e) : 0; }();
  // long rows and few of them -> 2 rows x 8 chunks per task (more waves); otherwise 4 rows x 4 chunks
  const bool two = forced ? (forced == 2) : (p.K >= 8192 || n_out <= 8192);
  const int rows = two ? 2 : 4;
  const int n_quads = (p.act == GV_ACT_SWIGLU) ? n_out / (rows / 2) : (n_out + rows - 1) / rows;
  int blocks = (n_quads + 3) / 4;   // one task row-group per wave until ~4096 waves, then several per wave
  if (blocks > 1024) blocks = 1024 + (blocks - 1024) / 8;
  if (blocks > 2048) blocks = 2048;
  vis_clear_error();
  const dim3 g(blocks), b(256);
  if (p.nb == 1) {
    if (two) hipLaunchKernelGGL((gemv_fp8w_kernel<2, 8, 1>), g, b, (size_t)p.K * 2, stream, p);
    else hipLaunchKernelGGL((gemv_fp8w_kernel<4, 4, 1>), g, b, (size_t)p.K * 2, stream, p);
    return vis_check_launch();
  }
  static const bool attr_ok = [] {
    return hipFuncSetAttribute((const void*)gemv_fp8w_kernel<2, 8, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, GV_ROWS_LDS_MAX) == hipSuccess &&
           hipFuncSetAttribute((const void*)gemv_fp8w_kernel<4, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, GV_ROWS_LDS_MAX) == hipSuccess &&
           hipFuncSetAttribute((const void*)gemv_fp8w_kernel<2, 8, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, GV_ROWS_LDS_MAX) == hipSuccess &&
           hipFuncSetAttribute((const void*)gemv_fp8w_kernel<4, 4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, GV_ROWS_LDS_MAX) == hipSuccess;
  }();
  if (!attr_ok) return VIS_ERR_LAUNCH;
  if (p.nb == 2) {
    if (two) hipLaunchKernelGGL((gemv_fp8w_kernel<2, 8, 2>), g, b, (size_t)p.K * 4, stream, p);
    else hipLaunchKernelGGL((gemv_fp8w_kernel<4, 4, 2>), g, b, (size_t)p.K * 4, stream, p);
  } else {
    if (two) hipLaunchKernelGGL((gemv_fp8w_kernel<2, 8, 4>), g, b, (size_t)p.K * 8, stream, p);
    else hipLaunchKernelGGL((gemv_fp8w_kernel<4, 4, 4>), g, b, (size_t)p.K * 8, stream, p);
  }
  return vis_check_launch();
}

extern "C" int vis_gemv_fp8w(const void* x, const void* Wq, const void* scale, const void* bias, const void* R,
                             const void* norm_w, void* y, int N, int K, int ldw, int act, int out_f32, float eps,
                             hipStream_t stream) {
  if (!x || !Wq || !scale || !y || N <= 0 || K <= 0) return VIS_ERR_ARG;
  if (K % 16 != 0 || ldw % 16 != 0 || ldw < K || K * 2 > 60 * 1024) return VIS_ERR_ARG;
  if (act != GV_ACT_NONE && act != GV_ACT_SWIGLU) return VIS_ERR_ARG;
  if (act == GV_ACT_SWIGLU && (N % 64 != 0 || bias || R || out_f32)) return VIS_ERR_ARG;
  if (((uintptr_t)x | (uintptr_t)Wq | (uintptr_t)norm_w) & 15) return VIS_ERR_ARG;
  if ((uintptr_t)scale & 3) return VIS_ERR_ARG;
  GemvF8Args p;
  p.x = (const bf16_t*)x; p.W = (const uint8_t*)Wq; p.scale = (const float*)scale; p.bias = (const bf16_t*)bias;
  p.R = (const bf16_t*)R; p.norm_w = (const bf16_t*)norm_w; p.y = y;
  p.N = N; p.K = K; p.ldw = ldw; p.act = act; p.out_f32 = out_f32; p.eps = eps;
  p.nb = 1; p.ldx = 0; p.ldy = 0; p.ldr = 0;
  return gemv_fp8w_launch(p, stream);
}

// vis_gemv_fp8w for B <= 4 input rows (W8A16: bf16 activations, e4m3 weights streamed once for all rows); each row's
// arithmetic is vis_gemv_fp8w's, so a handful of in-flight sequences decode bit-identically to one.
extern "C" int vis_gemv_fp8w_rows(const void* x, const void* Wq, const void* scale, const void* bias, const void* R,
                                  const void* norm_w, void* y, int B, int N, int K, int ldw, int ldx, int ldy, int ldr,
                                  int act, int out_f32, float eps, hipStream_t stream) {
  if (!x || !Wq || !scale || !y || N <= 0 || K <= 0 || B < 1 || B > 4) return VIS_ERR_ARG;
  if (K % 16 != 0 || ldw % 16 != 0 || ldw < K || ldx % 8 != 0 || ldx < K) return VIS_ERR_ARG;
  const int nbk = (B == 1) ? 1 : (B == 2 ? 2 : 4);
  if ((size_t)K * 2 * nbk > (size_t)(nbk == 1 ? 60 * 1024 : GV_ROWS_LDS_MAX)) return VIS_ERR_ARG;
  if (act != GV_ACT_NONE && act != GV_ACT_SWIGLU) return VIS_ERR_ARG;
  if (act == GV_ACT_SWIGLU && (N % 64 != 0 || bias || R || out_f32)) return VIS_ERR_ARG;
  if (ldy < ((act == GV_ACT_SWIGLU) ? N / 2 : N) || (R && ldr < N)) return VIS_ERR_ARG;
  if (((uintptr_t)x | (uintptr_t)Wq | (uintptr_t)norm_w) & 15) return VIS_ERR_ARG;
  if ((uintptr_t)scale & 3) return VIS_ERR_ARG;
  GemvF8Args p;
  p.x = (const bf16_t*)x; p.W = (const uint8_t*)Wq; p.scale = (const float*)scale; p.bias = (const bf16_t*)bias;
  p.R = (const bf16_t*)R; p.norm_w = (const bf16_t*)norm_w; p.y = y;
  p.N = N; p.K = K; p.ldw = ldw; p.act = act; p.out_f32 = out_f32; p.eps = eps;
  p.nb = B; p.ldx = ldx; p.ldy = ldy; p.ldr = ldr;
  return gemv_fp8w_launch(p, stream);
}

// (the split decode attention item - DecAttnArgs, decode_attn_split_body - lives in decode_common.hip.h)
template <int G>
__global__ __launch_bounds__(256) void decode_attn_fused_kernel(DecAttnArgs p) {
  __shared__ DecAttnLds<G> L;
  decode_attn_split_body<G, false>(p, L, blockIdx.x, blockIdx.y, blockIdx.z, ChainCtx{nullptr, nullptr, nullptr, 0u, nullptr});
}

// ---------------------------------------------------------------------------
// Batched form: ONE workgroup per (kv head, sequence) walks the whole context - no partials in HBM, no combine launch.
// The split form above needs Hkv x splits x batch workgroups and a second launch; at 64 in-flight sequences that is
// 16384 short workgroups per layer (each pays its own prologue: rope, q staging, one exposed load round trip) and the KV
// cache moves at 3.2 TB/s.  Here the 8 waves of a 512-thread workgroup are INDEPENDENT streams: wave w owns the 16-key
// steps j = w, w + 8, ... of its sequence, with its own online softmax per head and its own share of O in registers, so
// the loop has no barrier at all (a first version with one 64-key tile per workgroup iteration and two barriers per tile
// ran at 2.7 TB/s: four waves in lockstep cannot hide their own latencies).  Per step and wave:
//   loads   : 16 K rows in the MFMA A layout (4 x 16 B per lane) + 16 V rows (two dims per lane), two register sets
//             (static indexing): the next step's loads are issued before the current step is consumed;
//   scores  : S^T[key][head] on the MFMA; the 16-key max / sum per head are in-lane + two permlane swaps (no LDS);
//   P * V   : p[head][key] goes through a wave-private LDS strip (same-wave DS ops are ordered: no barrier) and comes
//             back as broadcast ds_read_b128; each lane accumulates its two output dims for all G heads.
// The eight waves' (m, l, O) are merged once at the end through LDS.  Same arithmetic per (head, key) as the split
// kernel (bf16-rounded rotated q / k, f32 softmax in the log2 domain, f32 P*V); the summation order over keys differs, so
// the two forms agree to f32 rounding, not bit for bit.
template <int G>
__global__ __launch_bounds__(512) void decode_attn_stream_kernel(DecAttnArgs p, bf16_t* __restrict__ out) {
  constexpr int HD = 128, HALF = 64, NW = 8;
  __shared__ __attribute__((aligned(16))) bf16_t q_s[16][HD];   // heads >= G are zero
  __shared__ __attribute__((aligned(16))) bf16_t knew_s[HD];
  __shared__ __attribute__((aligned(16))) bf16_t vnew_s[HD];
  __shared__ __attribute__((aligned(16))) uint32_t pbuf[NW][G][8];   // wave-private probabilities of the current step: bf16 pairs
                                                                     // (keys 2 j, 2 j + 1) - the B operand of v_dot2_f32_bf16
  __shared__ float pal[NW][G];                                     // wave-private rescale factors of the current step
  __shared__ __attribute__((aligned(16))) float red_o[NW][G][HD];
  __shared__ float red_ml[NW][G][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, h = lane >> 4;
  const int hkv = blockIdx.x, seq = blockIdx.y;
  const bf16_t* Kh0 = p.k_cache + (size_t)hkv * p.cache_tokens * 128;     // sequence 0's rows (the batch's shared prefix)
  const bf16_t* Vh0 = p.v_cache + (size_t)hkv * p.cache_tokens * 128;
  p.qkv += seq * p.qkv_bs;
  p.k_cache += seq * p.cache_bs;
  p.v_cache += seq * p.cache_bs;
  p.cos_t += seq * p.tab_bs;
  p.sin_t += seq * p.tab_bs;
  p.step_ptr += seq;
  out += (size_t)seq * p.Hq * HD;
  bf16_t* Kh = p.k_cache + (size_t)hkv * p.cache_tokens * HD;
  bf16_t* Vh = p.v_cache + (size_t)hkv * p.cache_tokens * HD;

  // V (r05): a lane holds EIGHT dims (8 l15 ..) of FOUR keys (4 h ..) as four 16-byte loads - an instruction moves four whole
  // 256-byte rows - instead of two dims of all sixteen keys as sixteen 4-byte loads (an instruction moved one row): the same
  // bytes and the same arithmetic per (head, key, dim) in a quarter of the V requests; the four key groups' partial sums meet once,
  // after the last step.  Measured (64 sequences, same box, three alternating pairs): 5.245 -> 5.228 ms per decode step - the launch
  // is not bound by its V request count either.
  struct StepRegs { u32x4 k[4]; u32x4 v[4]; };
  auto load_step = [&](StepRegs& r, int j) {          // keys 16 j .. 16 j + 15 (rows clamped: addresses are always valid)
    const int k0 = j * 16;
    const bool sh = k0 < p.shared_len;             // (wave-uniform; shared_len is a multiple of 16)
    const bf16_t* Kb = sh ? Kh0 : Kh;
    const bf16_t* Vb = sh ? Vh0 : Vh;
    const int krow = min(k0 + l15, p.cache_tokens - 1);
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) r.k[ds] = *(const u32x4*)(Kb + (size_t)krow * HD + ds * 32 + 8 * h);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = min(k0 + 4 * h + i, p.cache_tokens - 1);
      r.v[i] = *(const u32x4*)(Vb + (size_t)row * HD + 8 * l15);
    }
  };
  // Two register sets.  (Negative, r05: FOUR sets - three steps = 24 KB per wave in flight instead of one - on the theory that
  // 8 KB per wave at ~2 us of loaded latency caps a CU at ~32 GB/s: 49.1 -> 52.3 us per layer at 64 sequences, 226 VGPRs.  The
  // launch is not short of bytes in flight; r03's counters say the texture addresser spends 45 % of its cycles stalled by the
  // cache behind it: 16 half-line K requests and 16 two-line V requests per step and wave.)
  StepRegs r0, r1;
  load_step(r0, wave);                   // issued before the position has even arrived (fixed rows)

  const int slot = min(*p.step_ptr, p.cache_tokens - 1);
  const int ctx = slot + 1;
  const int nsteps = (ctx + 15) >> 4;
  const bool cross = p.q_norm_w != nullptr;  // workgroup-uniform
  if (!cross) {
    const float* cr = p.cos_t + (size_t)slot * HD;
    const float* sr = p.sin_t + (size_t)slot * HD;
    for (int it = tid; it < (G + 1) * HALF; it += 512) {
      const int g = it / HALF, d = it - g * HALF;
      const int head = (g < G) ? hkv * G + g : p.Hq + hkv;
      bf16_t qa, qb;
      da_qkv2(p, seq, head * HD + d, head * HD + HALF + d, qa, qb);
      const float a = bf2f(qa), b = bf2f(qb);
      const bf16_t oa = f2bf(a * cr[d] - b * sr[d]);
      const bf16_t ob = f2bf(b * cr[HALF + d] + a * sr[HALF + d]);
      bf16_t* dst = (g < G) ? q_s[g] : knew_s;
      dst[d] = oa;
      dst[HALF + d] = ob;
    }
    for (int it = tid; it < (16 - G) * HD; it += 512) q_s[G + it / HD][it % HD] = 0;
    if (tid < HD) vnew_s[tid] = da_qkv(p, seq, (p.Hq + p.Hkv + hkv) * HD + tid);
    __syncthreads();
    if (tid < HD) {  // KV-cache append: this workgroup is the only reader and writer of the row in this launch
      Kh[(size_t)slot * HD + tid] = knew_s[tid];
      Vh[(size_t)slot * HD + tid] = vnew_s[tid];
    }
  } else {
    for (int g = wave; g < G; g += NW) {
      const bf16_t* qh = p.qkv + (size_t)(hkv * G + g) * HD;
      const float a = bf2f(qh[lane]), b = bf2f(qh[lane + 64]);
      const float ss = wave_sum(a * a + b * b);
      const float rstd = rsqrtf(ss * (1.0f / HD) + p.q_eps);
      q_s[g][lane] = f2bf(bf2f(f2bf(a * rstd)) * bf2f(p.q_norm_w[lane]));
      q_s[g][lane + 64] = f2bf(bf2f(f2bf(b * rstd)) * bf2f(p.q_norm_w[lane + 64]));
    }
    for (int it = tid; it < (16 - G) * HD; it += 512) q_s[G + it / HD][it % HD] = 0;
    if (tid < HD) { knew_s[tid] = 0; vnew_s[tid] = 0; }
    __syncthreads();
  }
  const int new_row = cross ? -1 : slot;   // the cache row whose value may still be only in knew_s / vnew_s
  bf16x8 qf[4];
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) qf[ds] = *(const bf16x8*)(&q_s[l15][ds * 32 + 8 * h]);
  const u32x4 vnew = *(const u32x4*)(&vnew_s[8 * l15]);
  float oacc[G][8];                        // O[g][8 l15 + e] over this lane's keys (4 h .. 4 h + 3 of every step)
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int e = 0; e < 8; ++e) oacc[g][e] = 0.f;
  float m_run = -1.0e30f, l_run = 0.f;     // of head l15 (identical in the four lanes l15, l15 + 16, + 32, + 48)

  auto step = [&](StepRegs& r, int j) {
    const int k0 = j * 16;
    // ---- scores: acc[e] = S^T[key = k0 + 4 h + e][head = l15]
    const bool is_new = (k0 + l15 == new_row);
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) {
      u32x4 kv = r.k[ds];
      const u32x4 nv = *(const u32x4*)(&knew_s[ds * 32 + 8 * h]);
      if (is_new) kv = nv;
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kv), qf[ds], acc, 0, 0, 0);
    }
    float sv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) sv[e] = (k0 + 4 * h + e < ctx) ? acc[e] * p.scale_log2 : -1.0e30f;
    const float mx = hmax4(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])));
    const float m_new = fmaxf(m_run, mx);
    // (v_exp_f32 itself: libm's exp2f wraps it in a compare, two selects, an add and a v_ldexp for results below 2^-126 - 25 of the
    // step's ~250 vector instructions - which here are probabilities that round to zero weight either way; measured: no change
    // at 64 sequences, 5.255 vs 5.258 ms per step - the launch reads its K / V bytes at 6.1 TB/s and that is its bound)
    const float al = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    float pe[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) pe[e] = __builtin_amdgcn_exp2f(sv[e] - m_new);
    // r05: P * V on v_dot2_f32_bf16 - the probabilities are rounded to bf16 (as the prefill attention's P is; exact products,
    // f32 accumulation) and two keys go through one instruction: 112 dot2 + 16 v_perm per step and wave instead of 224 FMAs + 32
    // unpacking shifts / masks.  This kernel is bound by its instruction stream, not by memory: with every key of every sequence
    // served from L2 it ran 45 of 62 us (r03 probe), and four register sets in flight instead of two made it slower (above).
    // The denominator sums the ROUNDED values, so numerator and denominator stay consistent.
    const uint32_t p01 = pack2bf(pe[0], pe[1]), p23 = pack2bf(pe[2], pe[3]);
    l_run = l_run * al + hsum4((__uint_as_float(p01 << 16) + __uint_as_float(p01 & 0xffff0000u)) +
                               (__uint_as_float(p23 << 16) + __uint_as_float(p23 & 0xffff0000u)));
    // the rescale is skipped while no head's running maximum moved (wave-uniform; after the first steps it rarely does)
    const bool resc = __any((l15 < G) && (al != 1.0f));
    if (l15 < G) {
      *(u32x2*)(&pbuf[wave][l15][2 * h]) = (u32x2){p01, p23};     // keys 4 h .. 4 h + 3 = pairs 2 h, 2 h + 1
      if (h == 0) pal[wave][l15] = al;
    }
    // other LANES' probabilities are read next: the hardware keeps a wave's DS operations in order, the COMPILER must not move the
    // reads of rows this thread did not write above the writes (single-thread semantics would let it)
    asm volatile("" ::: "memory");
    // ---- O[g][d] = O[g][d] * alpha[g] + sum_key p[g][key] V[key][d]; lane owns d = 8 l15 .. + 7 over keys 4 h .. 4 h + 3
    bf16x2 va[2][8];     // [key pair (4 h + 2 q, 4 h + 2 q + 1)][dim 8 l15 + e]: that dim of both keys
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const u32x4 r0 = (k0 + 4 * h + 2 * q == new_row) ? vnew : r.v[2 * q];
      const u32x4 r1 = (k0 + 4 * h + 2 * q + 1 == new_row) ? vnew : r.v[2 * q + 1];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        va[q][2 * m] = __builtin_bit_cast(bf16x2, __builtin_amdgcn_perm(r1[m], r0[m], 0x05040100u));       // {r0.lo, r1.lo}
        va[q][2 * m + 1] = __builtin_bit_cast(bf16x2, __builtin_amdgcn_perm(r1[m], r0[m], 0x07060302u));   // {r0.hi, r1.hi}
      }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      // this lane's two key pairs of head g, as two scalar reads: the 8-byte vector read `u32x2 pq = *(const u32x2*)&pbuf[..][2 h]`
      // followed by pq[0] / pq[1] came out of hipcc 7.2 with BOTH dot products on element 0 (the IR holds a single extractelement:
      // tools/probes/decode_attn_forms.py shows a one-key context at exactly twice its value) - test_decode_attn_streaming_form caught it
      const uint32_t pw0 = pbuf[wave][g][2 * h], pw1 = pbuf[wave][g][2 * h + 1];
      const bf16x2 pp0 = __builtin_bit_cast(bf16x2, pw0), pp1 = __builtin_bit_cast(bf16x2, pw1);
      if (resc) {
        const float ag = pal[wave][g];
#pragma unroll
        for (int e = 0; e < 8; ++e) oacc[g][e] *= ag;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {       // keys past the context carry p = 0 (masked scores)
        const float a0 = __builtin_amdgcn_fdot2_f32_bf16(va[0][e], pp0, oacc[g][e], false);
        oacc[g][e] = __builtin_amdgcn_fdot2_f32_bf16(va[1][e], pp1, a0, false);
      }
    }
  };

  // wave w: steps w, w + 8, w + 16, ...; two register sets alternate (static indexing): the loads of the next step are
  // in flight while the current one is consumed, and two waves share each SIMD
  for (int j = wave; j < nsteps; j += 2 * NW) {
    if (j + NW < nsteps) load_step(r1, j + NW);
    step(r0, j);
    if (j + NW < nsteps) {
      if (j + 2 * NW < nsteps) load_step(r0, j + 2 * NW);
      step(r1, j + NW);
    }
  }

  // ---- merge the eight waves
#pragma unroll
  for (int g = 0; g < G; ++g) {
    float t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = hsum4(oacc[g][e]);           // the four key groups of the wave (lanes l15, + 16, + 32, + 48)
    if (h == 0) {
      *(f32x4*)(&red_o[wave][g][8 * l15]) = (f32x4){t[0], t[1], t[2], t[3]};
      *(f32x4*)(&red_o[wave][g][8 * l15 + 4]) = (f32x4){t[4], t[5], t[6], t[7]};
    }
  }
  if (h == 0 && l15 < G) { red_ml[wave][l15][0] = m_run; red_ml[wave][l15][1] = l_run; }
  __syncthreads();
  for (int i = tid; i < G * HD; i += 512) {
    const int g = i / HD, d = i - g * HD;
    float M = red_ml[0][g][0];
#pragma unroll
    for (int w = 1; w < NW; ++w) M = fmaxf(M, red_ml[w][g][0]);
    float o = 0.f, l = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float wt = exp2f(red_ml[w][g][0] - M);
      o += wt * red_o[w][g][d];
      l += wt * red_ml[w][g][1];
    }
    out[(size_t)(hkv * G + g) * HD + d] = f2bf(l > 0.f ? o / l : 0.f);
  }
}

// merge the per-split partials of one query head (only the splits that ran): 256 threads = 128 dims x 2
// split-halves; the first partial loads are issued before the position is known
#define CB_PRE 32   // partial rows preloaded per thread: covers 64 splits = 4096 cached keys
__global__ __launch_bounds__(256) void decode_attn_combine_kernel(const float* __restrict__ part_o,
                                                                  const float* __restrict__ part_ml,
                                                                  bf16_t* __restrict__ out, int nsplit,
                                                                  const int* __restrict__ step_ptr,
                                                                  int cache_tokens) {
  constexpr int HD = 128;
  __shared__ float wgt[256];
  __shared__ float osum[2][HD];
  __shared__ float wm[4];
  const int hq = blockIdx.x, seq = blockIdx.y, tid = threadIdx.x, d = tid & 127, half = tid >> 7;
  part_o += (size_t)seq * gridDim.x * nsplit * HD;
  part_ml += (size_t)seq * gridDim.x * nsplit * 2;
  out += (size_t)seq * gridDim.x * HD;
  step_ptr += seq;
  const float* po = part_o + (size_t)hq * nsplit * HD + d;
  const float* ml = part_ml + (size_t)hq * nsplit * 2;
  // ONE level of loads: the partials of the first 2 x CB_PRE split slots, the statistics of every slot and the position
  // are all requested at once; slots that did not run this step hold stale data and are masked (selected away, never
  // multiplied) once the position has arrived.  Before, position -> statistics -> late partials were three dependent
  // round trips in a kernel that does nothing else.
  float v[CB_PRE];
#pragma unroll
  for (int j = 0; j < CB_PRE; ++j) v[j] = po[(size_t)min(half + 2 * j, nsplit - 1) * HD];
  const int sl = min(tid, nsplit - 1);
  const float m_raw = ml[sl * 2], l_raw = ml[sl * 2 + 1];
  const int ctx = min(*step_ptr, cache_tokens - 1) + 1;
  const int active = min((ctx + DA_MAXKEYS - 1) / DA_MAXKEYS, nsplit);  // splits that ran
  const float m = (tid < active) ? m_raw : -1.0e30f;
  const float l = (tid < active) ? l_raw : 0.f;
  float M = wave_max(m);
  if ((tid & 63) == 0) wm[tid >> 6] = M;
  __syncthreads();
  M = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
  const float w = (tid < active) ? exp2f(m - M) : 0.f;
  wgt[tid] = w;
  const float lw = wave_sum(w * l);
  __syncthreads();
  if ((tid & 63) == 0) wm[tid >> 6] = lw;
  float o = 0.f;
#pragma unroll
  for (int j = 0; j < CB_PRE; ++j) {
    const int s = half + 2 * j;
    if (s < active) o += wgt[s] * v[j];
  }
  for (int s = half + 2 * CB_PRE; s < active; s += 2) o += wgt[s] * po[(size_t)s * HD];   // contexts past 64 splits
  osum[half][d] = o;
  __syncthreads();
  if (tid < HD) {
    const float lt = wm[0] + wm[1] + wm[2] + wm[3];
    const float ot = osum[0][tid] + osum[1][tid];
    out[(size_t)hq * HD + tid] = f2bf(lt > 0.f ? ot / lt : 0.f);
  }
}

static int decode_attn_launch(const DecAttnArgs& p, void* out, int batch, hipStream_t stream) {
  const int G = p.Hq / p.Hkv;
  vis_clear_error();
  // Enough (kv head, sequence) pairs to fill the chip on their own: the streaming form - one workgroup per pair, the
  // whole context in one pass, no partials, no combine.  VIS_DECODE_ATTN_STREAM=0 keeps the split form (A/B).
  // (read per launch - not cached - so that a test can compare both forms in one process; graph capture bakes it in)
  const char* se = getenv("VIS_DECODE_ATTN_STREAM");
  const int stream_env = se ? atoi(se) : 1;
  if (stream_env == 2 || (stream_env && p.Hkv * batch >= 128)) {   // 2: force (tests at small batch)
    const dim3 grid(p.Hkv, batch), block(512);
    switch (G) {
      case 1: hipLaunchKernelGGL(decode_attn_stream_kernel<1>, grid, block, 0, stream, p, (bf16_t*)out); break;
      case 2: hipLaunchKernelGGL(decode_attn_stream_kernel<2>, grid, block, 0, stream, p, (bf16_t*)out); break;
      case 4: hipLaunchKernelGGL(decode_attn_stream_kernel<4>, grid, block, 0, stream, p, (bf16_t*)out); break;
      case 7: hipLaunchKernelGGL(decode_attn_stream_kernel<7>, grid, block, 0, stream, p, (bf16_t*)out); break;
      default: hipLaunchKernelGGL(decode_attn_stream_kernel<8>, grid, block, 0, stream, p, (bf16_t*)out); break;
    }
    return vis_check_launch();
  }
  const dim3 grid(p.Hkv, p.nsplit, batch), block(256);
  switch (G) {
    case 1: hipLaunchKernelGGL(decode_attn_fused_kernel<1>, grid, block, 0, stream, p); break;
    case 2: hipLaunchKernelGGL(decode_attn_fused_kernel<2>, grid, block, 0, stream, p); break;
    case 4: hipLaunchKernelGGL(decode_attn_fused_kernel<4>, grid, block, 0, stream, p); break;
    case 7: hipLaunchKernelGGL(decode_attn_fused_kernel<7>, grid, block, 0, stream, p); break;
    default: hipLaunchKernelGGL(decode_attn_fused_kernel<8>, grid, block, 0, stream, p); break;
  }
  hipLaunchKernelGGL(decode_attn_combine_kernel, dim3(p.Hq, batch), dim3(256), 0, stream, (const float*)p.part_o,
                     (const float*)p.part_ml, (bf16_t*)out, p.nsplit, p.step_ptr, p.cache_tokens);
  return vis_check_launch();
}

extern "C" int vis_decode_attn(const void* qkv, const void* cos_t, const void* sin_t, void* k_cache, void* v_cache,
                               const void* step_ptr, void* part_o, void* part_ml, void* out, int Hq, int Hkv, int HD,
                               int cache_tokens, int nsplit, float scale, int batch, long long qkv_bs,
                               long long cache_bs, long long tab_bs, hipStream_t stream) {
  if (!qkv || !cos_t || !sin_t || !k_cache || !v_cache || !step_ptr || !part_o || !part_ml || !out)
    return VIS_ERR_ARG;
  if (batch <= 0 || batch > 64 || (batch > 1 && (qkv_bs <= 0 || cache_bs <= 0 || tab_bs < 0))) return VIS_ERR_ARG;
  if ((qkv_bs % 8) || (cache_bs % 8)) return VIS_ERR_ARG;
  if (HD != 128 || Hq <= 0 || Hkv <= 0 || Hq % Hkv != 0) return VIS_ERR_ARG;
  const int G = Hq / Hkv;
  if (G != 1 && G != 2 && G != 4 && G != 7 && G != 8) return VIS_ERR_ARG;  // instantiated GQA group sizes
  if (nsplit <= 0 || nsplit > 256 || cache_tokens <= 0) return VIS_ERR_ARG;
  if ((long long)nsplit * DA_MAXKEYS < cache_tokens) return VIS_ERR_ARG;  // every key must be covered
  if (((uintptr_t)k_cache | (uintptr_t)v_cache) & 15) return VIS_ERR_ARG;
  DecAttnArgs p;
  p.qkv = (const bf16_t*)qkv; p.cos_t = (const float*)cos_t; p.sin_t = (const float*)sin_t;
  p.k_cache = (bf16_t*)k_cache; p.v_cache = (bf16_t*)v_cache; p.step_ptr = (const int*)step_ptr;
  p.part_o = (float*)part_o; p.part_ml = (float*)part_ml;
  p.Hq = Hq; p.Hkv = Hkv; p.cache_tokens = cache_tokens; p.nsplit = nsplit;
  p.scale_log2 = scale * 1.4426950408889634f;
  p.qkv_bs = qkv_bs; p.cache_bs = cache_bs; p.tab_bs = tab_bs;
  p.q_norm_w = nullptr; p.q_eps = 0.f;
  p.shared_len = 0;
  da_no_parts(p);
  return decode_attn_launch(p, out, batch, stream);
}

// vis_decode_attn for a batch whose sequences share their first `shared_len` cached keys (a multiple of 64: the common
// text prefix of a batch inspection, which the prompt passes copied into every slot): those keys / values are read from
// sequence 0's cache by every sequence - one HBM read and L2 hits instead of `batch` HBM reads of identical rows.  Same
// values, same arithmetic: results are bit-identical to vis_decode_attn.  The new token's slot must lie beyond the shared
// range (it always does: the prefix is part of the prompt).  Streaming form only (Hkv * batch >= 128); smaller batches
// ignore shared_len.
extern "C" int vis_decode_attn_shared(const void* qkv, const void* cos_t, const void* sin_t, void* k_cache, void* v_cache,
                                      const void* step_ptr, void* part_o, void* part_ml, void* out, int Hq, int Hkv, int HD,
                                      int cache_tokens, int nsplit, float scale, int batch, long long qkv_bs,
                                      long long cache_bs, long long tab_bs, int shared_len, hipStream_t stream) {
  if (shared_len < 0 || shared_len % 64 != 0 || shared_len >= cache_tokens) return VIS_ERR_ARG;
  if (!qkv || !cos_t || !sin_t || !k_cache || !v_cache || !step_ptr || !part_o || !part_ml || !out)
    return VIS_ERR_ARG;
  if (batch <= 0 || batch > 64 || (batch > 1 && (qkv_bs <= 0 || cache_bs <= 0 || tab_bs < 0))) return VIS_ERR_ARG;
  if ((qkv_bs % 8) || (cache_bs % 8)) return VIS_ERR_ARG;
  if (HD != 128 || Hq <= 0 || Hkv <= 0 || Hq % Hkv != 0) return VIS_ERR_ARG;
  const int G = Hq / Hkv;
  if (G != 1 && G != 2 && G != 4 && G != 7 && G != 8) return VIS_ERR_ARG;
  if (nsplit <= 0 || nsplit > 256 || cache_tokens <= 0) return VIS_ERR_ARG;
  if ((long long)nsplit * DA_MAXKEYS < cache_tokens) return VIS_ERR_ARG;
  if (((uintptr_t)k_cache | (uintptr_t)v_cache) & 15) return VIS_ERR_ARG;
  DecAttnArgs p;
  p.qkv = (const bf16_t*)qkv; p.cos_t = (const float*)cos_t; p.sin_t = (const float*)sin_t;
  p.k_cache = (bf16_t*)k_cache; p.v_cache = (bf16_t*)v_cache; p.step_ptr = (const int*)step_ptr;
  p.part_o = (float*)part_o; p.part_ml = (float*)part_ml;
  p.Hq = Hq; p.Hkv = Hkv; p.cache_tokens = cache_tokens; p.nsplit = nsplit;
  p.scale_log2 = scale * 1.4426950408889634f;
  p.qkv_bs = qkv_bs; p.cache_bs = cache_bs; p.tab_bs = tab_bs;
  p.q_norm_w = nullptr; p.q_eps = 0.f;
  p.shared_len = shared_len;
  da_no_parts(p);
  return decode_attn_launch(p, out, batch, stream);
}

// vis_decode_attn_shared without the qkv finalisation launch in front of it: the batched qkv projection (vis_gemm_decode_bf16 /
// _fp8) leaves `ksplit` f32 partial slabs of `slab_rows` x n_qkv (slab_rows = 16 / 32 / 64 for batch <= 16 / <= 32 / beyond, as
// vis_skinny_finalize assumes); every (kv head, sequence) workgroup sums the 1152 columns it needs itself - fixed order, * sx[b] *
// sw[n] for fp8 partials, + bias, one rounding to bf16: skinny_finalize_kernel's arithmetic (fin_plain_value), so results are
// bit-identical to vis_skinny_finalize(..) + vis_decode_attn_shared(..), one launch and one ~5 us dependent kernel per layer less.
extern "C" int vis_decode_attn_parts(const void* part, int ksplit, int slab_rows, const void* bias, const void* sx,
                                     const void* sw, const void* cos_t, const void* sin_t, void* k_cache, void* v_cache,
                                     const void* step_ptr, void* part_o, void* part_ml, void* out, int Hq, int Hkv, int HD,
                                     int cache_tokens, int nsplit, float scale, int batch, long long cache_bs,
                                     long long tab_bs, int shared_len, hipStream_t stream) {
  if (!part || !cos_t || !sin_t || !k_cache || !v_cache || !step_ptr || !part_o || !part_ml || !out) return VIS_ERR_ARG;
  if (shared_len < 0 || shared_len % 64 != 0 || shared_len >= cache_tokens) return VIS_ERR_ARG;
  if (batch <= 0 || batch > 64 || (batch > 1 && (cache_bs <= 0 || tab_bs < 0)) || (cache_bs % 8)) return VIS_ERR_ARG;
  if (ksplit < 1 || ksplit > 16 || slab_rows < batch || (slab_rows != 16 && slab_rows != 32 && slab_rows != 64)) return VIS_ERR_ARG;
  if ((sx != nullptr) != (sw != nullptr)) return VIS_ERR_ARG;
  if (HD != 128 || Hq <= 0 || Hkv <= 0 || Hq % Hkv != 0) return VIS_ERR_ARG;
  const int G = Hq / Hkv;
  if (G != 1 && G != 2 && G != 4 && G != 7 && G != 8) return VIS_ERR_ARG;
  if (nsplit <= 0 || nsplit > 256 || cache_tokens <= 0 || (long long)nsplit * DA_MAXKEYS < cache_tokens) return VIS_ERR_ARG;
  if (((uintptr_t)k_cache | (uintptr_t)v_cache) & 15 || ((uintptr_t)part | (uintptr_t)sx | (uintptr_t)sw) & 3 || ((uintptr_t)bias & 1))
    return VIS_ERR_ARG;
  DecAttnArgs p;
  p.qkv = nullptr; p.cos_t = (const float*)cos_t; p.sin_t = (const float*)sin_t;
  p.k_cache = (bf16_t*)k_cache; p.v_cache = (bf16_t*)v_cache; p.step_ptr = (const int*)step_ptr;
  p.part_o = (float*)part_o; p.part_ml = (float*)part_ml;
  p.Hq = Hq; p.Hkv = Hkv; p.cache_tokens = cache_tokens; p.nsplit = nsplit;
  p.scale_log2 = scale * 1.4426950408889634f;
  p.qkv_bs = 0; p.cache_bs = cache_bs; p.tab_bs = tab_bs;
  p.q_norm_w = nullptr; p.q_eps = 0.f;
  p.shared_len = shared_len;
  p.part_n = (Hq + 2 * Hkv) * 128;
  p.qkv_part = (const float*)part; p.part_stride = (long long)slab_rows * p.part_n; p.part_ks = ksplit;
  p.qkv_bias = (const bf16_t*)bias; p.part_sx = (const float*)sx; p.part_sw = (const float*)sw;
  return decode_attn_launch(p, out, batch, stream);
}

// Cross-attention of one new token over a static key/value set (mllama cross layers, decode): q [Hq*128] straight
// from the q projection (q_norm applied inside), keys/values [Hkv][key_tokens][128] written once per request by
// the prefill; *nkeys_m1 (device int) = number of valid keys - 1.  Same split/combine machinery as vis_decode_attn.
static int decode_cross_attn_impl(const void* q, const void* q_norm_w, const void* k, const void* v,
                                  const void* nkeys_m1, void* part_o, void* part_ml, void* out, int Hq, int Hkv,
                                  int HD, int key_tokens, int nsplit, float scale, float eps, int batch, long long q_bs,
                                  long long kv_bs, hipStream_t stream) {
  if (!q || !q_norm_w || !k || !v || !nkeys_m1 || !part_o || !part_ml || !out) return VIS_ERR_ARG;
  if (HD != 128 || Hq <= 0 || Hkv <= 0 || Hq % Hkv != 0) return VIS_ERR_ARG;
  const int G = Hq / Hkv;
  if (G != 1 && G != 2 && G != 4 && G != 7 && G != 8) return VIS_ERR_ARG;
  if (nsplit <= 0 || nsplit > 256 || key_tokens <= 0 || (long long)nsplit * DA_MAXKEYS < key_tokens) return VIS_ERR_ARG;
  if (batch <= 0 || batch > 64 || (batch > 1 && (q_bs <= 0 || kv_bs <= 0)) || (q_bs % 8) || (kv_bs % 8)) return VIS_ERR_ARG;
  if (((uintptr_t)k | (uintptr_t)v) & 15) return VIS_ERR_ARG;
  DecAttnArgs p;
  p.qkv = (const bf16_t*)q; p.cos_t = nullptr; p.sin_t = nullptr;
  p.k_cache = (bf16_t*)k; p.v_cache = (bf16_t*)v; p.step_ptr = (const int*)nkeys_m1;
  p.part_o = (float*)part_o; p.part_ml = (float*)part_ml;
  p.Hq = Hq; p.Hkv = Hkv; p.cache_tokens = key_tokens; p.nsplit = nsplit;
  p.scale_log2 = scale * 1.4426950408889634f;
  p.qkv_bs = q_bs; p.cache_bs = kv_bs; p.tab_bs = 0;
  p.q_norm_w = (const bf16_t*)q_norm_w; p.q_eps = eps;
  p.shared_len = 0;
  da_no_parts(p);
  return decode_attn_launch(p, out, batch, stream);
}

extern "C" int vis_decode_cross_attn(const void* q, const void* q_norm_w, const void* k, const void* v,
                                     const void* nkeys_m1, void* part_o, void* part_ml, void* out, int Hq, int Hkv,
                                     int HD, int key_tokens, int nsplit, float scale, float eps, hipStream_t stream) {
  return decode_cross_attn_impl(q, q_norm_w, k, v, nkeys_m1, part_o, part_ml, out, Hq, Hkv, HD, key_tokens, nsplit, scale,
                                eps, 1, 0, 0, stream);
}

// Batch form (mllama batched decode): sequence b reads q + b * q_bs, its own static keys / values k, v + b * kv_bs
// (element strides) and nkeys_m1[b]; out [batch][Hq * 128]; workspaces sized batch x the single-sequence ones.
extern "C" int vis_decode_cross_attn_batch(const void* q, const void* q_norm_w, const void* k, const void* v,
                                           const void* nkeys_m1, void* part_o, void* part_ml, void* out, int Hq,
                                           int Hkv, int HD, int key_tokens, int nsplit, float scale, float eps,
                                           int batch, long long q_bs, long long kv_bs, hipStream_t stream) {
  return decode_cross_attn_impl(q, q_norm_w, k, v, nkeys_m1, part_o, part_ml, out, Hq, Hkv, HD, key_tokens, nsplit, scale,
                                eps, batch, q_bs, kv_bs, stream);
}

// ---------------------------------------------------------------------------
// Greedy pick.  Stage 1: per-block (max, first index); stage 2: one block merges,
// writes tokens[*step] = argmax, cur_token = argmax and then *step += 1.
// temperature sampling = Gumbel-max: argmax(logit/T + g_i), g_i = -log(-log(u_i)), u_i from a counter hash of
// (seed, step, i).  Exact categorical sampling, no softmax pass, no host round trip, graph-replayable.
__global__ __launch_bounds__(256) void argmax_stage1_kernel(const float* __restrict__ logits, int V,
                                                            float* __restrict__ bval, int* __restrict__ bidx,
                                                            float inv_temp, unsigned seed,
                                                            const int* __restrict__ step_ptr, int ld_logits) {
  const int tid = threadIdx.x, seq = blockIdx.y;
  logits += (size_t)seq * ld_logits;
  bval += seq * 256;
  bidx += seq * 256;
  seed += 0x9E3779B9u * (unsigned)seq;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  const unsigned step = (unsigned)step_ptr[seq];
  for (int i = blockIdx.x * 256 + tid; i < V; i += gridDim.x * 256) {
    float v = logits[i];
    if (inv_temp > 0.f) v = v * inv_temp + gumbel_noise(seed, step, (unsigned)i);
    if (v > best || (v == best && i < bi)) { best = v; bi = i; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  __shared__ float sv[4];
  __shared__ int si[4];
  if ((tid & 63) == 0) { sv[tid >> 6] = best; si[tid >> 6] = bi; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; ++w)
      if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
    bval[blockIdx.x] = best;
    bidx[blockIdx.x] = bi;
  }
}

__global__ __launch_bounds__(64) void argmax_stage2_kernel(const float* __restrict__ bval,
                                                           const int* __restrict__ bidx, int nb,
                                                           int* __restrict__ tokens, int max_tokens,
                                                           int* __restrict__ cur_token, int* __restrict__ step_ptr) {
  const int lane = threadIdx.x, seq = blockIdx.x;
  bval += seq * 256;
  bidx += seq * 256;
  tokens += (size_t)seq * max_tokens;
  cur_token += seq;
  step_ptr += seq;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int i = lane; i < nb; i += 64) {
    const float v = bval[i];
    const int ix = bidx[i];
    if (v > best || (v == best && ix < bi)) { best = v; bi = ix; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (lane == 0) {
    const int st = *step_ptr;
    if (st < max_tokens) tokens[st] = bi;
    *cur_token = bi;
    *step_ptr = st + 1;
  }
}

extern "C" int vis_argmax_f32(const void* logits, int V, void* ws_val, void* ws_idx, void* tokens, int max_tokens,
                              void* cur_token, void* step_ptr, float inv_temp, unsigned seed, int batch,
                              int ld_logits, hipStream_t stream) {
  if (!logits || V <= 0 || !ws_val || !ws_idx || !tokens || !cur_token || !step_ptr) return VIS_ERR_ARG;
  if (!(inv_temp >= 0.f) || batch <= 0 || batch > 64 || (batch > 1 && ld_logits < V)) return VIS_ERR_ARG;
  const int nb = min(256, (V + 255) / 256);
  vis_clear_error();
  hipLaunchKernelGGL(argmax_stage1_kernel, dim3(nb, batch), dim3(256), 0, stream, (const float*)logits, V,
                     (float*)ws_val, (int*)ws_idx, inv_temp, seed, (const int*)step_ptr, ld_logits);
  hipLaunchKernelGGL(argmax_stage2_kernel, dim3(batch), dim3(64), 0, stream, (const float*)ws_val, (const int*)ws_idx,
                     nb, (int*)tokens, max_tokens, (int*)cur_token, (int*)step_ptr);
  return vis_check_launch();
}

// argmax_stage2_kernel for up to 2048 per-workgroup maxima (vis_gemv_bf16_argmax): 256 threads, every thread's eight
// entries requested at once (the 64-thread form walks them in 32 dependent round trips: 11.5 us per token in the r04 trace)
__global__ __launch_bounds__(256) void argmax_merge_kernel(const float* __restrict__ bval, const int* __restrict__ bidx, int nb,
                                                           int* __restrict__ tokens, int max_tokens,
                                                           int* __restrict__ cur_token, int* __restrict__ step_ptr) {
  const int tid = threadIdx.x;
  float v[8];
  int ix[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int i = min(tid + 256 * k, nb - 1);        // (clamped duplicates lose every tie against themselves: harmless)
    v[k] = bval[i];
    ix[k] = bidx[i];
  }
  float best = -INFINITY;
  int bi = 0x7fffffff;
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (tid + 256 * k < nb && (v[k] > best || (v[k] == best && ix[k] < bi))) { best = v[k]; bi = ix[k]; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  __shared__ float sv[4];
  __shared__ int si[4];
  if ((tid & 63) == 0) { sv[tid >> 6] = best; si[tid >> 6] = bi; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; ++w)
      if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
    const int st = *step_ptr;
    if (st < max_tokens) tokens[st] = bi;
    *cur_token = bi;
    *step_ptr = st + 1;
  }
}

// K10 + K12 of the single-sequence step in two launches instead of three: logits = W rmsnorm(x) (f32, all N written, as
// vis_gemv_bf16 with out_f32) with the pick's first stage in the epilogue, then the merging launch of vis_argmax_f32
// (tokens[*step] = cur_token = pick, *step += 1).  Same comparison rule and the same Gumbel noise as vis_argmax_f32, so the
// pick is the one vis_gemv_bf16 + vis_argmax_f32 make (tests/test_kernels_gpu.py::test_gemv_argmax_equals_two_stage).
// ws_val / ws_idx: 2048 floats / ints.
extern "C" int vis_gemv_bf16_argmax(const void* x, const void* W, const void* norm_w, void* logits, int N, int K, int ldw,
                                    float eps, void* ws_val, void* ws_idx, void* tokens, int max_tokens, void* cur_token,
                                    void* step_ptr, float inv_temp, unsigned seed, hipStream_t stream) {
  if (!x || !W || !logits || !ws_val || !ws_idx || !tokens || !cur_token || !step_ptr || N <= 0 || K <= 0) return VIS_ERR_ARG;
  if (K % 8 != 0 || ldw % 8 != 0 || K * 2 > 60 * 1024 || !(inv_temp >= 0.f)) return VIS_ERR_ARG;
  if (((uintptr_t)x | (uintptr_t)W | (uintptr_t)norm_w) & 15) return VIS_ERR_ARG;
  GemvArgs p;
  p.x = (const bf16_t*)x; p.W = (const bf16_t*)W; p.bias = nullptr; p.R = nullptr; p.norm_w = (const bf16_t*)norm_w;
  p.y = logits; p.N = N; p.K = K; p.ldw = ldw; p.act = GV_ACT_NONE; p.out_f32 = 1; p.eps = eps; p.outs_per_block = 0;
  p.nb = 1; p.ldx = 0; p.ldy = 0; p.ldr = 0;
  p.am_val = (float*)ws_val; p.am_idx = (int*)ws_idx; p.am_step = (const int*)step_ptr; p.am_inv_temp = inv_temp;
  p.am_seed = seed;
  const int n_pairs = (N + 1) / 2;
  int blocks = (n_pairs + 3) / 4;                     // the grid rule of gemv_bf16_launch
  if (blocks > 1024) blocks = 1024 + (blocks - 1024) / 8;
  if (blocks > 2048) blocks = 2048;
  vis_clear_error();
  hipLaunchKernelGGL((gemv_bf16_kernel<1, true>), dim3(blocks), dim3(256), (size_t)K * 2, stream, p);
  hipLaunchKernelGGL(argmax_merge_kernel, dim3(1), dim3(256), 0, stream, (const float*)ws_val, (const int*)ws_idx, blocks,
                     (int*)tokens, max_tokens, (int*)cur_token, (int*)step_ptr);
  return vis_check_launch();
}
