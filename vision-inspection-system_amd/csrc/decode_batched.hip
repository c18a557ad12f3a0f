// K10 (batched): skinny GEMM for the decode step of up to 16 in-flight sequences on gfx950.
//
//   Y[b, n] = act( sum_k W[n, k] * xn[b, k] + bias[n] ) + R[b, n],   b < B <= 16
//   xn = x, or bf16(x * rstd[b]) * norm_w when norm_w != NULL (K3 fused; rstd[b] comes from the kernel that
//        produced x: vis_skinny_finalize / vis_rows_rstd)
// Small N (q/k/v, o, down: 28-36 row blocks) cannot fill 256 CUs, so K is additionally split over grid.y;
// split launches write f32 partials [ksplit][16][N] that vis_skinny_finalize sums in a fixed order (bitwise
// reproducible) while applying bias / residual and producing the next norm's statistics.
//
// One generated token per sequence needs every weight once, whatever B is: the kernel is HBM-bound on the
// weight stream exactly like the B = 1 GEMV (csrc/decode.hip), so batching B images multiplies images/s by ~B
// until prefill dominates ("batching decode across images is the main lever").
//
// MFMA formulation (v_mfma_f32_16x16x32_bf16): D[n][b] = W * xn^T with the WEIGHT rows as the A operand - a lane
// loads 16 bytes of its row (row = lane & 15, k = 8 * (lane >> 4) + j) straight from global memory into the
// operand registers (read-once data: no LDS round trip) - and xn^T as the B operand, read from an LDS copy of
// the current K-chunk of x (16 rows x 256 columns).  The accumulator holds 4 consecutive output features of
// one sequence per lane, so bias / residual / SwiGLU (16-row interleaved gate/up) are in-lane and the store is
// 8 bytes.  A wave owns two 16-row groups (32 weight rows = one SwiGLU group pair) and keeps both groups'
// accumulators across all K-chunks; the loads of the next K-chunk are issued before the current one is
// consumed (two register sets).
#include "common.hip.h"
#include <stdlib.h>

#define SK_KC 256          // K-chunk staged in LDS (8 k-steps of 32): two 64-VGPR weight buffers fit
#define SK_STEPS (SK_KC / 32)
#define SK_ROWS_PER_WAVE 32
#define SK_ROWS_PER_BLOCK (4 * SK_ROWS_PER_WAVE)

struct SkinnyArgs {
  const bf16_t* x;       // [B][ldx]
  const bf16_t* W;       // [N][ldw]
  const bf16_t* bias;    // [N] or null
  const bf16_t* R;       // [B][ldr] or null
  const bf16_t* norm_w;  // [K] or null
  const float* rstd;     // [B] (required with norm_w)
  float* part;           // [ksplit][16][N] f32 partials when ksplit > 1
  void* y;               // [B][ldy] bf16 or f32
  int B, N, K, ldx, ldw, ldr, ldy;
  int act, out_f32, ksplit;
};

struct SkBuf {
  u32x4 w[2][SK_STEPS];
};

template <bool NT>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(SkinnyArgs p) {
  // x chunk, row stride padded by 16 B so the 16 rows of a B-operand read land on distinct bank groups
  constexpr int XROW = SK_KC * 2 + 16;
  __shared__ __attribute__((aligned(16))) char xs[2][16 * XROW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, h = lane >> 4;
  const int row0 = blockIdx.x * SK_ROWS_PER_BLOCK + wave * SK_ROWS_PER_WAVE;
  const int nch_all = (p.K + SK_KC - 1) / SK_KC;
  const int ch0 = (int)((long long)nch_all * blockIdx.y / p.ksplit);       // this block's K-chunks [ch0, ch1)
  const int ch1 = (int)((long long)nch_all * (blockIdx.y + 1) / p.ksplit);
  const int nchunks = ch1 - ch0;
  const bool swiglu = (p.act == 3);

  // weight row pointers of this lane for the two 16-row groups (clamped: duplicates are never stored)
  const bf16_t* wrow[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) wrow[g] = p.W + (size_t)min(row0 + g * 16 + l15, p.N - 1) * p.ldw + 8 * h;

  auto load_w = [&](SkBuf& b, int chunk) {
    const int k0 = (ch0 + chunk) * SK_KC;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int s = 0; s < SK_STEPS; ++s) {
        const int k = min(k0 + s * 32, p.K - 32);  // unconditional load; out-of-range steps are zeroed via x
        b.w[g][s] = NT ? __builtin_nontemporal_load((const u32x4*)(wrow[g] + k)) : *(const u32x4*)(wrow[g] + k);
      }
  };

  SkBuf A, Bq;
  load_w(A, 0);  // in flight while the prologue runs

  // x staging is split: the global loads of chunk c+1 (x and norm_w: 2 x 16 B per thread) are issued FIRST,
  // then the 16 weight loads of chunk c+1, then chunk c is consumed, then the x registers are normalised and
  // written to LDS.  vmcnt retires in order, so waiting for the (older) x loads never drains the weight stream.
  constexpr int XCH = 16 * (SK_KC / 8);        // 512 16-byte chunks per x tile
  static_assert(XCH == 2 * 256, "two x chunks per thread");
  struct XRegs { u32x4 v[2], w[2]; };
  auto load_x = [&](XRegs& r, int chunk) {
    const int k0 = (ch0 + chunk) * SK_KC;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int it = tid + i * 256;
      const int b = it / (SK_KC / 8), c = it - b * (SK_KC / 8);
      const int k = min(k0 + c * 8, p.K - 8);
      r.v[i] = *(const u32x4*)(p.x + (size_t)min(b, p.B - 1) * p.ldx + k);
      r.w[i] = p.norm_w ? *(const u32x4*)(p.norm_w + k) : (u32x4){0u, 0u, 0u, 0u};
    }
  };
  auto store_x = [&](const XRegs& r, int buf, int chunk) {
    const int k0 = (ch0 + chunk) * SK_KC;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int it = tid + i * 256;
      const int b = it / (SK_KC / 8), c = it - b * (SK_KC / 8);
      u32x4 v = r.v[i];
      if (p.norm_w) {
        float f[8], w[8], o[8];
        unpack8(v, f);
        unpack8(r.w[i], w);
        const float rs = p.rstd[min(b, p.B - 1)];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = bf2f(f2bf(f[e] * rs)) * w[e];
        v = pack8(o);
      }
      if (b >= p.B || k0 + c * 8 >= p.K) v = (u32x4){0u, 0u, 0u, 0u};  // padding rows / columns contribute nothing
      *(u32x4*)(xs[buf] + b * XROW + c * 16) = v;
    }
  };

  f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
  auto consume = [&](const SkBuf& b, int buf) {
    const char* xb = xs[buf] + l15 * XROW + h * 16;
#pragma unroll
    for (int s = 0; s < SK_STEPS; ++s) {
      const bf16x8 xf = *(const bf16x8*)(xb + s * 64);
#pragma unroll
      for (int g = 0; g < 2; ++g)
        acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b.w[g][s]), xf, acc[g], 0, 0, 0);
    }
  };

  XRegs xr;
  load_x(xr, 0);
  store_x(xr, 0, 0);
  __syncthreads();
  for (int c = 0; c < nchunks; c += 2) {
    const bool more1 = (c + 1 < nchunks);
    if (more1) {
      load_x(xr, c + 1);
      load_w(Bq, c + 1);
    }
    consume(A, 0);
    if (more1) store_x(xr, 1, c + 1);
    __syncthreads();
    if (!more1) break;
    const bool more2 = (c + 2 < nchunks);
    if (more2) {
      load_x(xr, c + 2);
      load_w(A, c + 2);
    }
    consume(Bq, 1);
    if (more2) store_x(xr, 0, c + 2);
    __syncthreads();
  }

  // ---- epilogue: acc[g][r] = D[n = row0 + 16 g + 4 h + r][b = l15]
  const int b = l15;
  if (b >= p.B) return;
  if (p.ksplit > 1) {  // f32 partials; vis_skinny_finalize applies the epilogue
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int n = row0 + g * 16 + 4 * h;
      if (n < p.N) *(f32x4*)(p.part + ((size_t)blockIdx.y * 16 + b) * p.N + n) = acc[g];
    }
    return;
  }
  if (swiglu) {
    const int n = row0 + 4 * h;  // gate rows of this pair; up rows are +16
    if (n < p.N) {
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float g = acc[0][r], u = acc[1][r];
        v[r] = g / (1.0f + __expf(-g)) * u;
      }
      u32x2 o;
      o[0] = pack2bf(v[0], v[1]);
      o[1] = pack2bf(v[2], v[3]);
      *(u32x2*)((bf16_t*)p.y + (size_t)b * p.ldy + (row0 >> 1) + 4 * h) = o;
    }
    return;
  }
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int n = row0 + g * 16 + 4 * h;
    if (n >= p.N) continue;
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = acc[g][r];
    if (p.bias) {
      const u32x2 bb = *(const u32x2*)(p.bias + n);
      v[0] += __uint_as_float(bb[0] << 16);
      v[1] += __uint_as_float(bb[0] & 0xffff0000u);
      v[2] += __uint_as_float(bb[1] << 16);
      v[3] += __uint_as_float(bb[1] & 0xffff0000u);
    }
    if (p.R) {
      const u32x2 rr = *(const u32x2*)(p.R + (size_t)b * p.ldr + n);
      v[0] += __uint_as_float(rr[0] << 16);
      v[1] += __uint_as_float(rr[0] & 0xffff0000u);
      v[2] += __uint_as_float(rr[1] << 16);
      v[3] += __uint_as_float(rr[1] & 0xffff0000u);
    }
    if (p.out_f32) {
      *(f32x4*)((float*)p.y + (size_t)b * p.ldy + n) = (f32x4){v[0], v[1], v[2], v[3]};
    } else {
      u32x2 o;
      o[0] = pack2bf(v[0], v[1]);
      o[1] = pack2bf(v[2], v[3]);
      *(u32x2*)((bf16_t*)p.y + (size_t)b * p.ldy + n) = o;
    }
  }
}

// ksplit chosen so that (N / 128) * ksplit is at least ~256 workgroups
static int skinny_ksplit(int N, int K) {
  const int blocks = (N + SK_ROWS_PER_BLOCK - 1) / SK_ROWS_PER_BLOCK;
  const int nch = (K + SK_KC - 1) / SK_KC;
  int ks = (256 + blocks - 1) / blocks;
  if (ks > nch) ks = nch;
  if (ks > 16) ks = 16;
  return ks < 1 ? 1 : ks;
}

extern "C" int vis_skinny_ksplit(int N, int K) { return (N > 0 && K > 0) ? skinny_ksplit(N, K) : 0; }

extern "C" int vis_gemm_skinny_bf16(const void* x, const void* W, const void* bias, const void* R, const void* norm_w,
                                    const void* rstd, void* part, void* y, int B, int N, int K, int ldx, int ldw,
                                    int ldr, int ldy, int act, int out_f32, hipStream_t stream) {
  if (!x || !W || !y || B <= 0 || B > 16 || N <= 0 || K < 32) return VIS_ERR_ARG;
  if (K % 32 != 0 || N % 4 != 0 || ldx % 8 != 0 || ldw % 8 != 0 || ldy % 4 != 0 || (R && ldr % 4 != 0)) return VIS_ERR_ARG;
  if (act != 0 && act != 3) return VIS_ERR_ARG;
  if (act == 3 && (N % 32 != 0 || bias || R || out_f32)) return VIS_ERR_ARG;
  if (norm_w && !rstd) return VIS_ERR_ARG;
  if (((uintptr_t)x | (uintptr_t)W | (uintptr_t)norm_w | (uintptr_t)part) & 15) return VIS_ERR_ARG;
  if (((uintptr_t)y | (uintptr_t)bias | (uintptr_t)R) & 7) return VIS_ERR_ARG;
  if (out_f32 && ((uintptr_t)y & 15)) return VIS_ERR_ARG;
  SkinnyArgs p;
  p.x = (const bf16_t*)x; p.W = (const bf16_t*)W; p.bias = (const bf16_t*)bias; p.R = (const bf16_t*)R;
  p.norm_w = (const bf16_t*)norm_w; p.rstd = (const float*)rstd; p.part = (float*)part; p.y = y;
  p.B = B; p.N = N; p.K = K; p.ldx = ldx; p.ldw = ldw; p.ldr = ldr; p.ldy = ldy;
  p.act = act; p.out_f32 = out_f32;
  // SwiGLU and f32 logits are only produced by the direct (unsplit) epilogue; split launches need `part`
  p.ksplit = (act == 3 || out_f32 || !part) ? 1 : skinny_ksplit(N, K);
  const int blocks = (N + SK_ROWS_PER_BLOCK - 1) / SK_ROWS_PER_BLOCK;
  vis_clear_error();
  static const int nt = [] { const char* e = getenv("VIS_SKINNY_NT"); return e ? atoi(e) : 1; }();
  if (nt) hipLaunchKernelGGL(gemm_skinny_kernel<true>, dim3(blocks, p.ksplit), dim3(256), 0, stream, p);
  else hipLaunchKernelGGL(gemm_skinny_kernel<false>, dim3(blocks, p.ksplit), dim3(256), 0, stream, p);
  return vis_check_launch();
}

// ---------------------------------------------------------------------------
// vis_skinny_finalize: one workgroup per sequence.  y[b][n] = sum_ks part[ks][b][n] + bias[n] + R[b][n]
// (fixed summation order) and, when rstd_out != NULL, rstd_out[b] = rsqrt(mean(y[b]^2) + eps) over the bf16-
// rounded outputs - the statistics of the RMSNorm that the next fused skinny GEMM applies (TF:...:96-110).
// Called with ksplit == 0 it only computes rstd_out from the finished rows y (used after an unsplit launch).
__global__ __launch_bounds__(256) void skinny_finalize_kernel(const float* __restrict__ part, int ksplit,
                                                              const bf16_t* __restrict__ bias,
                                                              const bf16_t* __restrict__ R, bf16_t* __restrict__ y,
                                                              float* __restrict__ rstd_out, int N, int ldr, int ldy,
                                                              float eps) {
  const int b = blockIdx.x, tid = threadIdx.x;
  float ss = 0.f;
  for (int n = tid * 4; n < N; n += 1024) {
    float v[4];
    if (ksplit > 0) {
      f32x4 a = *(const f32x4*)(part + (size_t)b * N + n);
      for (int ks = 1; ks < ksplit; ++ks) {
        const f32x4 t = *(const f32x4*)(part + ((size_t)ks * 16 + b) * N + n);
        a[0] += t[0]; a[1] += t[1]; a[2] += t[2]; a[3] += t[3];
      }
      v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
      if (bias) {
        const u32x2 bb = *(const u32x2*)(bias + n);
        v[0] += __uint_as_float(bb[0] << 16); v[1] += __uint_as_float(bb[0] & 0xffff0000u);
        v[2] += __uint_as_float(bb[1] << 16); v[3] += __uint_as_float(bb[1] & 0xffff0000u);
      }
      if (R) {
        const u32x2 rr = *(const u32x2*)(R + (size_t)b * ldr + n);
        v[0] += __uint_as_float(rr[0] << 16); v[1] += __uint_as_float(rr[0] & 0xffff0000u);
        v[2] += __uint_as_float(rr[1] << 16); v[3] += __uint_as_float(rr[1] & 0xffff0000u);
      }
      u32x2 o;
      o[0] = pack2bf(v[0], v[1]);
      o[1] = pack2bf(v[2], v[3]);
      *(u32x2*)(y + (size_t)b * ldy + n) = o;
      v[0] = __uint_as_float(o[0] << 16); v[1] = __uint_as_float(o[0] & 0xffff0000u);
      v[2] = __uint_as_float(o[1] << 16); v[3] = __uint_as_float(o[1] & 0xffff0000u);
    } else {
      const u32x2 o = *(const u32x2*)(y + (size_t)b * ldy + n);
      v[0] = __uint_as_float(o[0] << 16); v[1] = __uint_as_float(o[0] & 0xffff0000u);
      v[2] = __uint_as_float(o[1] << 16); v[3] = __uint_as_float(o[1] & 0xffff0000u);
    }
    ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  if (!rstd_out) return;
  __shared__ float red[4];
  ss = wave_sum(ss);
  if ((tid & 63) == 0) red[tid >> 6] = ss;
  __syncthreads();
  if (tid == 0) rstd_out[b] = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)N + eps);
}

extern "C" int vis_skinny_finalize(const void* part, int ksplit, const void* bias, const void* R, void* y,
                                   void* rstd_out, int B, int N, int ldr, int ldy, float eps, hipStream_t stream) {
  if (!y || B <= 0 || B > 16 || N <= 0 || N % 4 != 0 || ksplit < 0 || ksplit > 16) return VIS_ERR_ARG;
  if (ksplit > 0 && !part) return VIS_ERR_ARG;
  if (ldy % 4 != 0 || (R && ldr % 4 != 0)) return VIS_ERR_ARG;
  if (((uintptr_t)part & 15) || (((uintptr_t)y | (uintptr_t)bias | (uintptr_t)R) & 7)) return VIS_ERR_ARG;
  vis_clear_error();
  hipLaunchKernelGGL(skinny_finalize_kernel, dim3(B), dim3(256), 0, stream, (const float*)part, ksplit,
                     (const bf16_t*)bias, (const bf16_t*)R, (bf16_t*)y, (float*)rstd_out, N, ldr, ldy, eps);
  return vis_check_launch();
}
