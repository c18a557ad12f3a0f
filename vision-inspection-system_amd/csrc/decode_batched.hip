// K10 (batched decode), second half: row-wise finalisation of the split-K projection of up to 32 in-flight
// sequences.  The first half is gemm_bf16_32x128_kernel (csrc/gemm_bf16.hip), which streams the weight panel
// once for all sequences and leaves f32 partial sums part[ks][16][N].
//
//   v[b][n]  = sum_ks part[ks][b][n] (fixed order: bitwise reproducible) + bias[n]
//   SwiGLU   : y[b][o] = silu(gate) * up over the 16-column interleaved gate/up layout (N/2 outputs)
//   residual : y = v + R[b][n]
//   next norm: yn[b][n] = bf16(bf16(y * rsqrt(mean(y^2) + eps)) * norm_w[n])   (K3 of the NEXT projection,
//              TF:models/qwen2_vl/modeling_qwen2_vl.py:96-110) - written in the same launch because one
//              workgroup owns a whole row.
// One 1024-thread workgroup per sequence, 4 consecutive columns per thread per pass, all partial loads of a
// pass issued before the adds (a first version with 256 threads and dependent loads took 12 us; this one ~4).
#include "common.hip.h"

#define FIN_THREADS 1024
#define FIN_MAXKS 16

struct FinArgs {
  const float* part;     // [ksplit][16 or 32][N] (32 rows per slab when B > 16)
  const bf16_t* bias;    // [N] or null
  const bf16_t* R;       // [B][ldr] or null
  const bf16_t* norm_w;  // [n_out] or null -> yn written
  bf16_t* y;             // [B][ldy]
  bf16_t* yn;            // [B][ldyn] or null
  int ksplit, N, ldr, ldy, ldyn, swiglu;
  float eps;
  // fp8 form (BASELINE configs[4]): partials are raw sums of e4m3 products -> v *= sx[b] * sw[n] first;
  // yq != null: the row handed to the NEXT projection (yn when normed, else y) is also written as e4m3 bytes
  // with its own per-row scale (amax / 448) - the activation quantiser fused here because this workgroup owns the row
  const float* sx;       // [B] or null
  const float* sw;       // [N] or null
  uint8_t* yq;           // [B][ldyq] or null
  float* yq_scale;       // [B]
  int ldyq;
};

template <int KS>
__device__ __forceinline__ f32x4 fin_sum(const float* base, size_t stride) {
  f32x4 t[KS];
#pragma unroll
  for (int k = 0; k < KS; ++k) t[k] = *(const f32x4*)(base + k * stride);
  f32x4 a = t[0];
#pragma unroll
  for (int k = 1; k < KS; ++k) { a[0] += t[k][0]; a[1] += t[k][1]; a[2] += t[k][2]; a[3] += t[k][3]; }
  return a;
}

__device__ __forceinline__ f32x4 fin_sum_dyn(const float* base, size_t stride, int ks) {
  switch (ks) {
    case 1: return fin_sum<1>(base, stride);
    case 2: return fin_sum<2>(base, stride);
    case 3: return fin_sum<3>(base, stride);
    case 4: return fin_sum<4>(base, stride);
    case 5: return fin_sum<5>(base, stride);
    case 6: return fin_sum<6>(base, stride);
    case 7: return fin_sum<7>(base, stride);
    case 8: return fin_sum<8>(base, stride);
    default: {
      f32x4 a = fin_sum<8>(base, stride);
      for (int k = 8; k < ks; ++k) {
        const f32x4 t = *(const f32x4*)(base + k * stride);
        a[0] += t[0]; a[1] += t[1]; a[2] += t[2]; a[3] += t[3];
      }
      return a;
    }
  }
}

__global__ __launch_bounds__(FIN_THREADS) void skinny_finalize_kernel(FinArgs p) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const size_t stride = (size_t)(gridDim.x > 32 ? 64 : gridDim.x > 16 ? 32 : 16) * p.N;  // a slab has 16 / 32 / 64 rows
  const float* pb = p.part + (size_t)b * p.N;
  const int n_out = p.swiglu ? (p.N >> 1) : p.N;
  float ss = 0.f;
  // pass over the OUTPUT columns, 4 per thread
  for (int o = tid * 4; o < n_out; o += FIN_THREADS * 4) {
    float v[4];
    if (p.swiglu) {
      const int g = ((o >> 4) << 5) + (o & 15);  // gate columns; the matching up columns are +16
      f32x4 ga = fin_sum_dyn(pb + g, stride, p.ksplit);
      f32x4 ua = fin_sum_dyn(pb + g + 16, stride, p.ksplit);
      if (p.sx) {
        const float sxb = p.sx[b];
        const f32x4 sg = *(const f32x4*)(p.sw + g), su = *(const f32x4*)(p.sw + g + 16);
#pragma unroll
        for (int r = 0; r < 4; ++r) { ga[r] *= sxb * sg[r]; ua[r] *= sxb * su[r]; }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = silu_fast(ga[r]) * ua[r];
    } else {
      const f32x4 a = fin_sum_dyn(pb + o, stride, p.ksplit);
      const bool scaled = p.sx != nullptr, has_b = p.bias != nullptr;
      const float sxb = scaled ? p.sx[b] : 0.f;
      const f32x4 s4 = scaled ? *(const f32x4*)(p.sw + o) : (f32x4){0.f, 0.f, 0.f, 0.f};
      const u32x2 bb = has_b ? *(const u32x2*)(p.bias + o) : (u32x2){0u, 0u};
      const float bf[4] = {__uint_as_float(bb[0] << 16), __uint_as_float(bb[0] & 0xffff0000u), __uint_as_float(bb[1] << 16),
                           __uint_as_float(bb[1] & 0xffff0000u)};
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = fin_plain_value(a[r], scaled, sxb, s4[r], has_b, bf[r]);   // (common.hip.h)
      if (p.R) {
        const u32x2 rr = *(const u32x2*)(p.R + (size_t)b * p.ldr + o);
        v[0] += __uint_as_float(rr[0] << 16); v[1] += __uint_as_float(rr[0] & 0xffff0000u);
        v[2] += __uint_as_float(rr[1] << 16); v[3] += __uint_as_float(rr[1] & 0xffff0000u);
      }
    }
    u32x2 q;
    q[0] = pack2bf(v[0], v[1]);
    q[1] = pack2bf(v[2], v[3]);
    *(u32x2*)(p.y + (size_t)b * p.ldy + o) = q;
    if (p.yn) {
      const float r0 = __uint_as_float(q[0] << 16), r1 = __uint_as_float(q[0] & 0xffff0000u);
      const float r2 = __uint_as_float(q[1] << 16), r3 = __uint_as_float(q[1] & 0xffff0000u);
      ss += r0 * r0 + r1 * r1 + r2 * r2 + r3 * r3;
    }
  }
  __shared__ float red[FIN_THREADS / 64];
  if (p.yn) {
    ss = wave_sum(ss);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < FIN_THREADS / 64; ++i) tot += red[i];
    const float rstd = rsqrtf(tot / (float)n_out + p.eps);
    for (int o = tid * 4; o < n_out; o += FIN_THREADS * 4) {  // this thread re-reads its own (bf16) outputs
      const u32x2 q = *(const u32x2*)(p.y + (size_t)b * p.ldy + o);
      const u32x2 w = *(const u32x2*)(p.norm_w + o);
      float f[4] = {__uint_as_float(q[0] << 16), __uint_as_float(q[0] & 0xffff0000u), __uint_as_float(q[1] << 16),
                    __uint_as_float(q[1] & 0xffff0000u)};
      const float g[4] = {__uint_as_float(w[0] << 16), __uint_as_float(w[0] & 0xffff0000u), __uint_as_float(w[1] << 16),
                          __uint_as_float(w[1] & 0xffff0000u)};
#pragma unroll
      for (int r = 0; r < 4; ++r) f[r] = bf2f(f2bf(f[r] * rstd)) * g[r];
      u32x2 t;
      t[0] = pack2bf(f[0], f[1]);
      t[1] = pack2bf(f[2], f[3]);
      *(u32x2*)(p.yn + (size_t)b * p.ldyn + o) = t;
    }
  }
  if (!p.yq) return;
  // ---- e4m3 copy of the row the next projection consumes (yn if normed, else y); own writes, re-read as bf16
  const bf16_t* src = p.yn ? p.yn + (size_t)b * p.ldyn : p.y + (size_t)b * p.ldy;
  float amax = 0.f;
  for (int o = tid * 4; o < n_out; o += FIN_THREADS * 4) {
    const u32x2 q = *(const u32x2*)(src + o);
    amax = fmaxf(fmaxf(amax, fabsf(__uint_as_float(q[0] << 16))), fabsf(__uint_as_float(q[0] & 0xffff0000u)));
    amax = fmaxf(fmaxf(amax, fabsf(__uint_as_float(q[1] << 16))), fabsf(__uint_as_float(q[1] & 0xffff0000u)));
  }
  amax = wave_max(amax);
  __syncthreads();   // red[] may still be read by slower waves of the norm pass
  if ((tid & 63) == 0) red[tid >> 6] = amax;
  __syncthreads();
  float am = 0.f;
#pragma unroll
  for (int i = 0; i < FIN_THREADS / 64; ++i) am = fmaxf(am, red[i]);
  const float sc = fmaxf(am / 448.0f, 1e-12f);
  const float inv = 1.0f / sc;
  if (tid == 0) p.yq_scale[b] = sc;
  for (int o = tid * 4; o < n_out; o += FIN_THREADS * 4) {
    const u32x2 q = *(const u32x2*)(src + o);
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(__uint_as_float(q[0] << 16) * inv, __uint_as_float(q[0] & 0xffff0000u) * inv, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(__uint_as_float(q[1] << 16) * inv, __uint_as_float(q[1] & 0xffff0000u) * inv, w, true);
    *(uint32_t*)(p.yq + (size_t)b * p.ldyq + o) = (uint32_t)w;
  }
}

static int skinny_finalize_launch(FinArgs p, int B, hipStream_t stream) {
  vis_clear_error();
  hipLaunchKernelGGL(skinny_finalize_kernel, dim3(B), dim3(FIN_THREADS), 0, stream, p);
  return vis_check_launch();
}

// fp8 form: partials of vis_gemm_decode_fp8 (raw) scaled by sx[b] * sw[n]; optional e4m3 copy (yq, yq_scale) of the
// row the next projection consumes (yn when norm_w is given, else y).  Everything else as vis_skinny_finalize.
extern "C" int vis_skinny_finalize_fp8(const void* part, int ksplit, const void* sx, const void* sw, const void* bias,
                                       const void* R, const void* norm_w, void* y, void* yn, void* yq, void* yq_scale,
                                       int B, int N, int ldr, int ldy, int ldyn, int ldyq, int swiglu, float eps,
                                       hipStream_t stream) {
  if (!part || !y || B <= 0 || B > 64 || N <= 0 || N % 4 != 0 || ksplit < 1 || ksplit > FIN_MAXKS) return VIS_ERR_ARG;
  if (swiglu && (N % 32 != 0 || bias || R)) return VIS_ERR_ARG;
  if ((yn != nullptr) != (norm_w != nullptr)) return VIS_ERR_ARG;
  if ((sx != nullptr) != (sw != nullptr) || (yq != nullptr) != (yq_scale != nullptr)) return VIS_ERR_ARG;
  if (ldy % 4 != 0 || (R && ldr % 4 != 0) || (yn && ldyn % 4 != 0) || (yq && ldyq % 4 != 0)) return VIS_ERR_ARG;
  if (((uintptr_t)part | (uintptr_t)sw) & 15 || ((uintptr_t)yq & 3) ||
      (((uintptr_t)y | (uintptr_t)yn | (uintptr_t)bias | (uintptr_t)R | (uintptr_t)norm_w) & 7))
    return VIS_ERR_ARG;
  FinArgs p;
  p.part = (const float*)part; p.bias = (const bf16_t*)bias; p.R = (const bf16_t*)R; p.norm_w = (const bf16_t*)norm_w;
  p.y = (bf16_t*)y; p.yn = (bf16_t*)yn;
  p.ksplit = ksplit; p.N = N; p.ldr = ldr; p.ldy = ldy; p.ldyn = ldyn; p.swiglu = swiglu; p.eps = eps;
  p.sx = (const float*)sx; p.sw = (const float*)sw; p.yq = (uint8_t*)yq; p.yq_scale = (float*)yq_scale; p.ldyq = ldyq;
  return skinny_finalize_launch(p, B, stream);
}

extern "C" int vis_skinny_finalize(const void* part, int ksplit, const void* bias, const void* R, const void* norm_w,
                                   void* y, void* yn, int B, int N, int ldr, int ldy, int ldyn, int swiglu,
                                   float eps, hipStream_t stream) {
  if (!part || !y || B <= 0 || B > 64 || N <= 0 || N % 4 != 0 || ksplit < 1 || ksplit > FIN_MAXKS) return VIS_ERR_ARG;
  if (swiglu && (N % 32 != 0 || bias || R)) return VIS_ERR_ARG;
  if ((yn != nullptr) != (norm_w != nullptr)) return VIS_ERR_ARG;
  if (ldy % 4 != 0 || (R && ldr % 4 != 0) || (yn && ldyn % 4 != 0)) return VIS_ERR_ARG;
  if (((uintptr_t)part & 15) || (((uintptr_t)y | (uintptr_t)yn | (uintptr_t)bias | (uintptr_t)R | (uintptr_t)norm_w) & 7))
    return VIS_ERR_ARG;
  FinArgs p;
  p.part = (const float*)part; p.bias = (const bf16_t*)bias; p.R = (const bf16_t*)R; p.norm_w = (const bf16_t*)norm_w;
  p.y = (bf16_t*)y; p.yn = (bf16_t*)yn;
  p.ksplit = ksplit; p.N = N; p.ldr = ldr; p.ldy = ldy; p.ldyn = ldyn; p.swiglu = swiglu; p.eps = eps;
  p.sx = nullptr; p.sw = nullptr; p.yq = nullptr; p.yq_scale = nullptr; p.ldyq = 0;
  return skinny_finalize_launch(p, B, stream);
}
