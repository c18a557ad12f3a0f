// K3 RMSNorm (TF:models/qwen2_vl/modeling_qwen2_vl.py:96-110) and K5 LayerNorm
// (ViT norm1/norm2 :428-429, merger ln_q :281) for gfx950.
//
// HBM-bound row kernels: one 64-lane wave per row, 16-byte (8 x bf16) loads,
// the row is held in registers between the reduction and the normalise pass
// so each element is read from HBM exactly once.  f32 statistics, bf16 I/O.
// Algorithmic bytes: 2*N*2 B per row (+ the affine vectors, L2-resident).
#include "common.hip.h"

#define NORM_MAX_CHUNKS 10  // per lane: supports N <= 64*8*10 = 5120

// r05: every load of a row is UNCONDITIONAL (clamped chunk / row index, contributions masked afterwards)
// and issued before the first use - the r01-r04 kernels guarded each 16-byte load with `if (c < nch)`, which makes hipcc drain
// vmcnt at every guard (the loads of a row went out one round trip at a time: 2.7 - 4.6 TB/s).  CH = chunks per lane is a
// template parameter (1280 columns: 3, 3584: 7), ROWS rows per wave keep more bytes in flight on short rows.
template <bool LAYERNORM, int CH, int ROWS>
__global__ __launch_bounds__(256) void norm_rows_kernel(const bf16_t* __restrict__ x,
                                                        const bf16_t* __restrict__ w,
                                                        const bf16_t* __restrict__ b,
                                                        bf16_t* __restrict__ y, int rows, int N,
                                                        int ldx, int ldy, float eps) {
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * ROWS;
  if (row0 >= rows) return;
  const int nch = N >> 3;  // 16-byte chunks per row
  u32x4 raw[ROWS][CH], wraw[CH], braw[CH];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const bf16_t* xr = x + (size_t)min(row0 + r, rows - 1) * ldx;
#pragma unroll
    for (int i = 0; i < CH; ++i) raw[r][i] = *(const u32x4*)(xr + min(lane + i * 64, nch - 1) * 8);
  }
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    wraw[i] = *(const u32x4*)(w + min(lane + i * 64, nch - 1) * 8);
    if (LAYERNORM) braw[i] = *(const u32x4*)(b + min(lane + i * 64, nch - 1) * 8);
  }
  const float inv_n = 1.0f / (float)N;
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    float v[CH][8];
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      unpack8(raw[r][i], v[i]);
      if (lane + i * 64 < nch) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          s += v[i][e];
          ss += v[i][e] * v[i][e];
        }
      }
    }
    s = wave_sum(s);
    ss = wave_sum(ss);
    float mean = 0.f, rstd;
    if (LAYERNORM) {
      mean = s * inv_n;
      float d2 = 0.f;   // two-pass variance from registers (no cancellation)
#pragma unroll
      for (int i = 0; i < CH; ++i)
        if (lane + i * 64 < nch) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float d = v[i][e] - mean;
            d2 += d * d;
          }
        }
      d2 = wave_sum(d2);
      rstd = rsqrtf(d2 * inv_n + eps);
    } else {
      rstd = rsqrtf(ss * inv_n + eps);
    }
    if (row0 + r >= rows) continue;   // (wave-uniform)
    bf16_t* yr = y + (size_t)(row0 + r) * ldy;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int c = lane + i * 64;
      float wv[8], o[8];
      unpack8(wraw[i], wv);
      if (LAYERNORM) {
        float bv[8];
        unpack8(braw[i], bv);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mean) * rstd * wv[e] + bv[e];
      } else {
        // HF rounds the normalised value to the input dtype before the weight multiply
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = bf2f(f2bf(v[i][e] * rstd)) * wv[e];
      }
      if (c < nch) *(u32x4*)(yr + c * 8) = pack8(o);
    }
  }
}

// Split-K finalisation fused with the NEXT norm (the row owner exists here: one wave per row):
//   x[m] = bf16( sum_s part[s][m] + bias + R[m] )      (fixed slice order: bitwise reproducible)
//   y[m] = norm(x[m]) * w (+ b)                         (statistics of the ROUNDED x, exactly what a separate
//                                                         vis_rmsnorm_bf16 / vis_layernorm_bf16 of x would compute)
// replaces gemm_splitk_finalize_kernel + norm_rows_kernel: one pass over the partials instead of a pass over the
// partials and a read-modify-write pass over x.  y == NULL: finalisation only.  KS = slices (2: every use in the engines;
// 0: run-time count), CH = chunks per lane; all loads of the row issued up front (see norm_rows_kernel).
template <bool LAYERNORM, int CH, int KS>
__global__ __launch_bounds__(256) void finalize_norm_rows_kernel(const float* __restrict__ part, int ksplit, size_t slice,
                                                                 const bf16_t* __restrict__ bias,
                                                                 const bf16_t* __restrict__ R, int ldr,
                                                                 bf16_t* __restrict__ xo, int ldxo,
                                                                 const bf16_t* __restrict__ w,
                                                                 const bf16_t* __restrict__ b, bf16_t* __restrict__ y,
                                                                 int ldy, int rows, int N, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nch = N >> 3;
  const float* pr = part + (size_t)row * N;
  constexpr int KSL = KS > 0 ? KS : 1;
  f32x4 plo[KSL][CH], phi[KSL][CH];
  u32x4 braw[CH], rraw[CH], wraw[CH], nbraw[CH];
  const u32x4 z = (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int c = min(lane + i * 64, nch - 1);
#pragma unroll
    for (int k = 0; k < KSL; ++k) {
      plo[k][i] = *(const f32x4*)(pr + k * slice + c * 8);
      phi[k][i] = *(const f32x4*)(pr + k * slice + c * 8 + 4);
    }
    braw[i] = bias ? *(const u32x4*)(bias + c * 8) : z;                    // (pointer tests are kernel-uniform)
    rraw[i] = R ? *(const u32x4*)(R + (size_t)row * ldr + c * 8) : z;
    if (y) {
      wraw[i] = *(const u32x4*)(w + c * 8);
      nbraw[i] = LAYERNORM ? *(const u32x4*)(b + c * 8) : z;
    }
  }
  float v[CH][8];
  float s = 0.f, ss = 0.f;
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int c = lane + i * 64;
    float a[8], f[8];
    *(f32x4*)a = plo[0][i];
    *(f32x4*)(a + 4) = phi[0][i];
#pragma unroll
    for (int k = 1; k < KSL; ++k) {
      a[0] += plo[k][i][0]; a[1] += plo[k][i][1]; a[2] += plo[k][i][2]; a[3] += plo[k][i][3];
      a[4] += phi[k][i][0]; a[5] += phi[k][i][1]; a[6] += phi[k][i][2]; a[7] += phi[k][i][3];
    }
    if (KS == 0) {
      for (int ks = 1; ks < ksplit; ++ks) {
        const int cc = min(c, nch - 1);
        const f32x4 lo = *(const f32x4*)(pr + ks * slice + cc * 8), hi = *(const f32x4*)(pr + ks * slice + cc * 8 + 4);
        a[0] += lo[0]; a[1] += lo[1]; a[2] += lo[2]; a[3] += lo[3];
        a[4] += hi[0]; a[5] += hi[1]; a[6] += hi[2]; a[7] += hi[3];
      }
    }
    unpack8(braw[i], f);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] += f[e];      // (+ 0 without a bias: a + 0.0f == a)
    unpack8(rraw[i], f);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] += f[e];
    const u32x4 packed = pack8(a);
    if (c < nch) *(u32x4*)(xo + (size_t)row * ldxo + c * 8) = packed;
    unpack8(packed, v[i]);                    // the norm sees the rounded values
    if (c < nch) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        s += v[i][e];
        ss += v[i][e] * v[i][e];
      }
    }
  }
  if (!y) return;
  s = wave_sum(s);
  ss = wave_sum(ss);
  const float inv_n = 1.0f / (float)N;
  float mean = 0.f, rstd;
  if (LAYERNORM) {
    mean = s * inv_n;
    float d2 = 0.f;
#pragma unroll
    for (int i = 0; i < CH; ++i)
      if (lane + i * 64 < nch) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = v[i][e] - mean;
          d2 += d * d;
        }
      }
    d2 = wave_sum(d2);
    rstd = rsqrtf(d2 * inv_n + eps);
  } else {
    rstd = rsqrtf(ss * inv_n + eps);
  }
  bf16_t* yr = y + (size_t)row * ldy;
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int c = lane + i * 64;
    float wv[8], o[8];
    unpack8(wraw[i], wv);
    if (LAYERNORM) {
      float bv[8];
      unpack8(nbraw[i], bv);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mean) * rstd * wv[e] + bv[e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = bf2f(f2bf(v[i][e] * rstd)) * wv[e];
    }
    if (c < nch) *(u32x4*)(yr + c * 8) = pack8(o);
  }
}

// Rows wider than 3 chunks per lane (LLM hidden 3584: 7) keep the r02 form: loads per chunk inside the loop, up to ten chunks of
// registers.  The all-loads-up-front form above needs 242 VGPRs at 7 chunks x 2 slices (two waves per SIMD) and measured
// SLOWER there (cold-cache microbench tools/rowpass_bench.py: 43.9 vs 39.7 us at 2249 x 3584, 28.7 vs 24.9 at 1289 rows), while
// the 1280-column ViT rows gain (30.6 vs 32.6 us): occupancy, not the number of loads in flight per wave, is what these row
// passes live on.
template <bool LAYERNORM>
__global__ __launch_bounds__(256) void finalize_norm_rows_wide_kernel(const float* __restrict__ part, int ksplit, size_t slice,
                                                                 const bf16_t* __restrict__ bias,
                                                                 const bf16_t* __restrict__ R, int ldr,
                                                                 bf16_t* __restrict__ xo, int ldxo,
                                                                 const bf16_t* __restrict__ w,
                                                                 const bf16_t* __restrict__ b, bf16_t* __restrict__ y,
                                                                 int ldy, int rows, int N, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nch = N >> 3;
  const float* pr = part + (size_t)row * N;
  float v[NORM_MAX_CHUNKS][8];
  float s = 0.f, ss = 0.f;
#pragma unroll
  for (int i = 0; i < NORM_MAX_CHUNKS; ++i) {
    const int c = lane + i * 64;
    if (c < nch) {
      float a[8];
      *(f32x4*)a = *(const f32x4*)(pr + c * 8);
      *(f32x4*)(a + 4) = *(const f32x4*)(pr + c * 8 + 4);
      for (int ks = 1; ks < ksplit; ++ks) {
        const f32x4 lo = *(const f32x4*)(pr + ks * slice + c * 8), hi = *(const f32x4*)(pr + ks * slice + c * 8 + 4);
        a[0] += lo[0]; a[1] += lo[1]; a[2] += lo[2]; a[3] += lo[3];
        a[4] += hi[0]; a[5] += hi[1]; a[6] += hi[2]; a[7] += hi[3];
      }
      if (bias) {
        float f[8];
        unpack8(*(const u32x4*)(bias + c * 8), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] += f[e];
      }
      if (R) {
        float f[8];
        unpack8(*(const u32x4*)(R + (size_t)row * ldr + c * 8), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] += f[e];
      }
      const u32x4 packed = pack8(a);
      *(u32x4*)(xo + (size_t)row * ldxo + c * 8) = packed;
      unpack8(packed, v[i]);                    // the norm sees the rounded values
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        s += v[i][e];
        ss += v[i][e] * v[i][e];
      }
    }
  }
  if (!y) return;
  s = wave_sum(s);
  ss = wave_sum(ss);
  const float inv_n = 1.0f / (float)N;
  float mean = 0.f, rstd;
  if (LAYERNORM) {
    mean = s * inv_n;
    float d2 = 0.f;
#pragma unroll
    for (int i = 0; i < NORM_MAX_CHUNKS; ++i) {
      const int c = lane + i * 64;
      if (c < nch) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = v[i][e] - mean;
          d2 += d * d;
        }
      }
    }
    d2 = wave_sum(d2);
    rstd = rsqrtf(d2 * inv_n + eps);
  } else {
    rstd = rsqrtf(ss * inv_n + eps);
  }
  bf16_t* yr = y + (size_t)row * ldy;
#pragma unroll
  for (int i = 0; i < NORM_MAX_CHUNKS; ++i) {
    const int c = lane + i * 64;
    if (c < nch) {
      float wv[8], o[8];
      unpack8(*(const u32x4*)(w + c * 8), wv);
      if (LAYERNORM) {
        float bv[8];
        unpack8(*(const u32x4*)(b + c * 8), bv);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mean) * rstd * wv[e] + bv[e];
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = bf2f(f2bf(v[i][e] * rstd)) * wv[e];
      }
      *(u32x4*)(yr + c * 8) = pack8(o);
    }
  }
}

// part [ksplit][M][N] f32 (vis_gemm_bf16_splitk_part) -> x [M][ldx] bf16 = sum + bias + R, and (y != NULL)
// y [M][ldy] = RMSNorm (norm_b == NULL) or LayerNorm of x.
extern "C" int vis_splitk_finalize_norm(const void* part, int ksplit, const void* bias, const void* R, void* x,
                                        const void* norm_w, const void* norm_b, void* y, int M, int N, int ldr, int ldx,
                                        int ldy, float eps, hipStream_t stream) {
  if (!part || !x || M <= 0 || N <= 0 || ksplit < 1 || ksplit > 8) return VIS_ERR_ARG;
  if (N % 8 != 0 || N > 64 * 8 * NORM_MAX_CHUNKS || ldx % 8 != 0 || (R && ldr % 8 != 0)) return VIS_ERR_ARG;
  if (y && (!norm_w || ldy % 8 != 0)) return VIS_ERR_ARG;
  if (((uintptr_t)part | (uintptr_t)bias | (uintptr_t)R | (uintptr_t)x | (uintptr_t)norm_w | (uintptr_t)norm_b |
       (uintptr_t)y) & 15)
    return VIS_ERR_ARG;
  const dim3 grid((M + 3) / 4), block(256);
  const size_t slice = (size_t)M * N;
  const int ch = (N / 8 + 63) / 64;   // chunks per lane
  vis_clear_error();
#define FIN_LAUNCH(LN, CH, KS)                                                                                              \
  hipLaunchKernelGGL((finalize_norm_rows_kernel<LN, CH, KS>), grid, block, 0, stream, (const float*)part, ksplit, slice,        \
                     (const bf16_t*)bias, (const bf16_t*)R, ldr, (bf16_t*)x, ldx, (const bf16_t*)norm_w, (const bf16_t*)norm_b, \
                     (bf16_t*)y, ldy, M, N, eps)
#define FIN_PICK(LN)                                                        \
  do {                                                                      \
    if (ksplit == 2 && ch <= 3) FIN_LAUNCH(LN, 3, 2);                       \
    else hipLaunchKernelGGL(finalize_norm_rows_wide_kernel<LN>, grid, block, 0, stream, (const float*)part, ksplit, slice,   \
                            (const bf16_t*)bias, (const bf16_t*)R, ldr, (bf16_t*)x, ldx, (const bf16_t*)norm_w,              \
                            (const bf16_t*)norm_b, (bf16_t*)y, ldy, M, N, eps);                                              \
  } while (0)
  if (norm_b) FIN_PICK(true);
  else FIN_PICK(false);
#undef FIN_PICK
#undef FIN_LAUNCH
  return vis_check_launch();
}

static int norm_launch(bool ln, const void* x, const void* w, const void* b, void* y, int rows, int N,
                       int ldx, int ldy, float eps, hipStream_t stream) {
  if (!x || !w || !y || rows <= 0 || N <= 0) return VIS_ERR_ARG;
  if (N % 8 != 0 || N > 64 * 8 * NORM_MAX_CHUNKS || ldx % 8 != 0 || ldy % 8 != 0) return VIS_ERR_ARG;
  if (ln && !b) return VIS_ERR_ARG;
  if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)y | (uintptr_t)b) & 15) return VIS_ERR_ARG;
  const int ch = (N / 8 + 63) / 64;   // chunks per lane
  const int rpw = ch <= 3 ? 2 : 1;    // rows per wave
  const dim3 grid((rows + 4 * rpw - 1) / (4 * rpw)), block(256);
  vis_clear_error();
#define NORM_LAUNCH(LN, CH, ROWS)                                                                                         \
  hipLaunchKernelGGL((norm_rows_kernel<LN, CH, ROWS>), grid, block, 0, stream, (const bf16_t*)x, (const bf16_t*)w,          \
                     (const bf16_t*)b, (bf16_t*)y, rows, N, ldx, ldy, eps)
#define NORM_PICK(LN)                                  \
  do {                                                 \
    if (ch <= 3) NORM_LAUNCH(LN, 3, 2);                \
    else if (ch <= 7) NORM_LAUNCH(LN, 7, 1);           \
    else NORM_LAUNCH(LN, NORM_MAX_CHUNKS, 1);          \
  } while (0)
  if (ln) NORM_PICK(true);
  else NORM_PICK(false);
#undef NORM_PICK
#undef NORM_LAUNCH
  return vis_check_launch();
}

// RMSNorm over every 128-wide head of a packed row (mllama's k_norm on the cross-attention keys, TF:models/mllama/modeling_mllama.py:
// 411-440): x [tokens][ldx] holds `heads` consecutive 128-element heads per token; one wave per token, a head per 16-lane DPP row,
// four heads per pass.  Head h of token t gets exactly what vis_rmsnorm_bf16 on the [tokens, 128] slice x[:, 128 h ..] computes - the
// same lane <-> chunk map, and wave_sum's first four DPP steps ARE the 16-lane row sum (its last step adds three rows of zeros there) -
// in one launch instead of `heads`.
__device__ __forceinline__ float row16_sum(float v) {
  v += VIS_DPP(v, 0xB1);   // quad_perm [1,0,3,2]
  v += VIS_DPP(v, 0x4E);   // quad_perm [2,3,0,1]
  v += VIS_DPP(v, 0x141);  // row_half_mirror
  v += VIS_DPP(v, 0x140);  // row_mirror
  return v;
}

__global__ __launch_bounds__(256) void rmsnorm_heads_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                            bf16_t* __restrict__ y, int tokens, int heads, int ldx, int ldy,
                                                            float eps) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= tokens) return;
  const int c = lane & 15, hrow = lane >> 4;
  float wv[8];
  unpack8(*(const u32x4*)(w + c * 8), wv);
  const bf16_t* xr = x + (size_t)t * ldx + c * 8;
  bf16_t* yr = y + (size_t)t * ldy + c * 8;
  for (int h0 = 0; h0 < heads; h0 += 4) {
    const int h = min(h0 + hrow, heads - 1);
    float v[8];
    unpack8(*(const u32x4*)(xr + h * 128), v);
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) ss += v[e] * v[e];
    ss = row16_sum(ss);
    const float rstd = rsqrtf(ss * (1.0f / 128.0f) + eps);
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = bf2f(f2bf(v[e] * rstd)) * wv[e];
    if (h0 + hrow < heads) *(u32x4*)(yr + h * 128) = pack8(o);
  }
}

extern "C" int vis_rmsnorm_heads_bf16(const void* x, const void* w, void* y, int tokens, int heads, int head_dim, int ldx,
                                      int ldy, float eps, hipStream_t stream) {
  if (!x || !w || !y || tokens <= 0 || heads <= 0 || head_dim != 128) return VIS_ERR_ARG;
  if (ldx % 8 != 0 || ldy % 8 != 0 || ldx < heads * 128 || ldy < heads * 128) return VIS_ERR_ARG;
  if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15) return VIS_ERR_ARG;
  vis_clear_error();
  hipLaunchKernelGGL(rmsnorm_heads_kernel, dim3((tokens + 3) / 4), dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)w,
                     (bf16_t*)y, tokens, heads, ldx, ldy, eps);
  return vis_check_launch();
}

extern "C" int vis_rmsnorm_bf16(const void* x, const void* w, void* y, int rows, int N, int ldx, int ldy,
                                float eps, hipStream_t stream) {
  return norm_launch(false, x, w, nullptr, y, rows, N, ldx, ldy, eps, stream);
}

extern "C" int vis_layernorm_bf16(const void* x, const void* w, const void* b, void* y, int rows, int N,
                                  int ldx, int ldy, float eps, hipStream_t stream) {
  return norm_launch(true, x, w, b, y, rows, N, ldx, ldy, eps, stream);
}
