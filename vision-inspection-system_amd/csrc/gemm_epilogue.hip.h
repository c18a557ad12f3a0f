// Wide (LDS-staged, 16 bytes per lane) GEMM epilogue shared by the bf16 and the fp8 tile kernels: both issue their
// MFMAs with W as the A operand and so hold the same accumulator layout.  `Args` needs bias, R, C, M, N, ldc, ldr, act, nt.
#pragma once
#include "common.hip.h"


enum { ACT_NONE = 0, ACT_QUICKGELU = 1, ACT_GELU_ERF = 2, ACT_SWIGLU = 3 };

// erf-GELU, 0.5 x (1 + erf(x / sqrt 2)) (TF "gelu": the Qwen2-VL merger and every MLP of the mllama vision tower).  r01-r04 called
// libm's erff out of line (~60 instructions per element): the Auditor's tower fc1 - 131.7 M outputs per four images - spent ~200 of its
// 438 us in this epilogue (r05: 0.77 PFLOP/s where the same shape with QuickGELU ran 1.14).  Now Abramowitz & Stegun 7.1.26 in the
// erfc form: e = poly(t) exp(-z^2), t = 1 / (1 + p z), z = |x| / sqrt 2, |erfc error| <= 1.5e-7, and
//   x >= 0: x (1 - e / 2)      x < 0: x e / 2        (no 1 - erf cancellation on the negative side)
// = 11 VALU + v_rcp_f32 + v_exp_f32.  Against the f64 function over [-12, 12]: max abs error 6.4e-7 (torch's own f32 gelu: 1.3e-6),
// bf16-rounded results differ in 5e-5 of the cases where |gelu| > 0.01, by one bf16 ulp (torch f32: 5.9e-5).
__device__ __forceinline__ float gelu_erf(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
  const float p = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float e = p * __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);
  return x * (x >= 0.f ? 1.0f - 0.5f * e : 0.5f * e);
}

__device__ __forceinline__ float act_apply(float x, int act) {
  if (act == ACT_QUICKGELU) return quickgelu_fast(x);
  if (act == ACT_GELU_ERF) return gelu_erf(x);
  return x;
}

// ---------------------------------------------------------------------------
// Wide epilogue: the wave's accumulator tile goes through a wave-private 16 KiB LDS region and leaves as whole
// 16-byte chunks of C rows (8 lanes cover one 128-byte row segment), instead of 8-byte stores at a row stride
// (every store instruction of the direct form touches 16 different rows in 32-byte pieces: tools/gemm_kscan.py put
// the fixed cost of a 256 x 256 tile round at 16.4 us, most of it this store tail).  The residual is read the same
// way (16 bytes per lane) and added in f32 BEFORE the one rounding to bf16, so results are bit-identical to the
// direct epilogue; bias and activation are applied in the accumulator layout (4 consecutive columns per lane).
//   plain / bias / act      : bf16 staging, 128-byte rows, chunk c of row r at slot c ^ ((r >> 1) & 7)
//   residual                : f32 staging in 64-row halves, 256-byte rows, chunk c at slot c ^ (r & 15)
//   SwiGLU (gate/up pairs)  : bf16 staging of the 8 NT output columns, 64-byte rows, slot c ^ ((r >> 2) & 3)
// (slot maps chosen so that the ds_write_b64 / b128 lane groups and the ds_read_b128 lane groups hit distinct banks).
// DS operations of one wave execute in order, so no barrier is needed between a wave's writes and its own reads.
// Preconditions (checked on the host, p.wide): N % 8 == 0, ldc % 8 == 0, ldr % 8 == 0, C / R 16-byte aligned.
// the C stores of the wide epilogue (timing probes swap the policy: tools/probes/gemm_probe.sh)
#if defined(GEMM_PROBE) && GEMM_PROBE == 3
#define EPI_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
#elif defined(GEMM_PROBE) && GEMM_PROBE == 4
#define EPI_STORE(ptr, val) do { if (p.M < 0) *(ptr) = (val); } while (0)
#else
#define EPI_STORE(ptr, val) do { if (p.nt) __builtin_nontemporal_store((val), (ptr)); else *(ptr) = (val); } while (0)
#endif
// `sc(v, i, j)`: what the accumulator block (i, j) is multiplied by on its way out (fp8: row scale x column scales)
struct EpiNoScale {
  __device__ __forceinline__ f32x4 operator()(const f32x4& v, int, int) const { return v; }
};
template <int ACT, int MI, int NT, class Args, class Scale = EpiNoScale>
__device__ __forceinline__ void gemm_epilogue_wide(const Args& p, f32x4 (&acc)[MI][NT], int mbase, int nbase,
                                                   int lane, char* st, const Scale& sc = Scale()) {
  const int l15 = lane & 15, h = lane >> 4;
  // Bias of the lane's four consecutive columns in each 16-column block: ONE 8-byte load per block, all NT issued
  // together before anything else (r02 timeline probe: a load + wait inside every (row block, column block) step
  // was 32 dependent L2 round trips = 4 us of a 7 us epilogue).  Columns beyond N are never stored: the address is
  // clamped, the value unused.  Without a bias the adds are skipped by a uniform branch around the whole loop nest.
  f32x4 bia[NT];
  const bool has_b = p.bias != nullptr;
  if (has_b) {
    u32x2 raw[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) raw[j] = *(const u32x2*)(p.bias + min(nbase + j * 16 + 4 * h, p.N - 4));
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      bia[j][0] = __uint_as_float(raw[j][0] << 16); bia[j][1] = __uint_as_float(raw[j][0] & 0xffff0000u);
      bia[j][2] = __uint_as_float(raw[j][1] << 16); bia[j][3] = __uint_as_float(raw[j][1] & 0xffff0000u);
    }
  } else {
#pragma unroll
    for (int j = 0; j < NT; ++j) bia[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  if constexpr (ACT == ACT_SWIGLU) {
    static_assert(NT % 2 == 0, "gate/up pairs");
    // staging: 64-byte rows (the 8 NT output columns), chunk c of row r at slot c ^ ((r >> 2) & 3); row = i * 16 + l15
    int woff[NT / 2];
#pragma unroll
    for (int j = 0; j < NT; j += 2)
      woff[j >> 1] = l15 * 64 + ((((j >> 1) * 2 + (h >> 1)) ^ ((l15 >> 2) & 3)) << 4) + (h & 1) * 8;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int j = 0; j < NT; j += 2) {   // interleaved like the weight rows: gate block j, up block j + 1
        float v[4];
        const f32x4 g = sc(acc[i][j], i, j), u = sc(acc[i][j + 1], i, j + 1);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          v[r] = has_b ? silu_fast(g[r] + bia[j][r]) * (u[r] + bia[j + 1][r]) : silu_fast(g[r]) * u[r];
        u32x2 o;
        o[0] = pack2bf(v[0], v[1]);
        o[1] = pack2bf(v[2], v[3]);
        *(u32x2*)(st + i * 1024 + woff[j >> 1]) = o;
      }
    }
    const int rr = lane >> 2, c = lane & 3;                          // 16 rows x 4 chunks per pass
    const int oc = (nbase >> 1) + c * 8;
    const int roff = rr * 64 + ((c ^ ((rr >> 2) & 3)) << 4);         // (it * 16 + rr) >> 2 & 3 == (rr >> 2) & 3
    const int rows_ok = (c < NT && oc * 2 < p.N) ? p.M - mbase - rr : 0;   // row it * 16 + rr is valid iff it * 16 < rows_ok
    bf16_t* cp = p.C + (size_t)(mbase + rr) * p.ldc + oc;
    const size_t cstep = (size_t)16 * p.ldc;
#pragma unroll
    for (int it = 0; it < MI; ++it) {
      const u32x4 o = *(const u32x4*)(st + it * 1024 + roff);
      if (it * 16 < rows_ok) EPI_STORE((u32x4*)(cp + it * cstep), o);
    }
    return;
  } else {
    const bool has_r = p.R != nullptr;
    if (!has_r) {
      // staging: 128-byte rows, chunk c of row r at slot c ^ ((r >> 1) & 7); row = i * 16 + l15 -> (r >> 1) & 7 = l15 >> 1
      int woff[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) woff[j] = l15 * 128 + (((j * 2 + (h >> 1)) ^ ((l15 >> 1) & 7)) << 4) + (h & 1) * 8;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          f32x4 v = sc(acc[i][j], i, j);
          if (has_b) v += bia[j];
          if constexpr (ACT != ACT_NONE) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = act_apply(v[r], ACT);
          }
          u32x2 o;
          o[0] = pack2bf(v[0], v[1]);
          o[1] = pack2bf(v[2], v[3]);
          *(u32x2*)(st + i * 2048 + woff[j]) = o;
        }
      }
      const int rr = lane >> 3, c = lane & 7;                        // 8 rows x 8 chunks per pass
      const int n = nbase + c * 8;
      // row it * 8 + rr: ((row >> 1) & 7) = (it * 4 + (rr >> 1)) & 7 -> two slot patterns (it even / odd)
      const int roff0 = rr * 128 + ((c ^ ((rr >> 1) & 7)) << 4), roff1 = rr * 128 + ((c ^ ((4 + (rr >> 1)) & 7)) << 4);
      const int rows_ok = (c < 2 * NT && n < p.N) ? p.M - mbase - rr : 0;
      bf16_t* cp = p.C + (size_t)(mbase + rr) * p.ldc + n;
      const size_t cstep = (size_t)8 * p.ldc;
#pragma unroll
      for (int it = 0; it < 2 * MI; ++it) {
        const u32x4 o = *(const u32x4*)(st + it * 1024 + ((it & 1) ? roff1 : roff0));
        if (it * 8 < rows_ok) EPI_STORE((u32x4*)(cp + it * cstep), o);
      }
      return;
    }
    // residual: f32 staging, 64 rows at a time (256-byte rows, chunk c at slot c ^ (r & 15)); the residual rows of a
    // half are requested (8 x 16 bytes per lane) BEFORE the half's accumulators are staged, so their latency is paid once
    const int rr = lane >> 3, g = lane & 7;
    const int n = nbase + g * 8;
    const int rows_ok = (g < 2 * NT && n < p.N) ? p.M - mbase - rr : 0;
    int woff[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) woff[j] = l15 * 256 + (((j * 4 + h) ^ l15) << 4);
#pragma unroll
    for (int hm = 0; hm < MI / 4; ++hm) {
      u32x4 res[8];
      const bf16_t* rp = p.R + (size_t)(mbase + hm * 64 + rr) * p.ldr + n;
#pragma unroll
      for (int it = 0; it < 8; ++it)
        res[it] = (hm * 64 + it * 8 < rows_ok) ? *(const u32x4*)(rp + (size_t)it * 8 * p.ldr) : (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          f32x4 v = sc(acc[hm * 4 + i][j], hm * 4 + i, j);
          if (has_b) v += bia[j];
          if constexpr (ACT != ACT_NONE) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = act_apply(v[r], ACT);
          }
          *(f32x4*)(st + i * 4096 + woff[j]) = v;
        }
      }
      bf16_t* cp = p.C + (size_t)(mbase + hm * 64 + rr) * p.ldc + n;
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int row = it * 8 + rr;
        const f32x4 a = *(const f32x4*)(st + row * 256 + (((2 * g) ^ (row & 15)) << 4));
        const f32x4 b = *(const f32x4*)(st + row * 256 + (((2 * g + 1) ^ (row & 15)) << 4));
        float f[8];
        unpack8(res[it], f);
        float v[8] = {a[0] + f[0], a[1] + f[1], a[2] + f[2], a[3] + f[3], b[0] + f[4], b[1] + f[5], b[2] + f[6], b[3] + f[7]};
        if (hm * 64 + it * 8 < rows_ok) EPI_STORE((u32x4*)(cp + (size_t)it * 8 * p.ldc), pack8(v));
      }
    }
  }
}

// run-time activation -> compile-time epilogue instance (keeps every instance branch-free and small)
template <int MI, int NT, class Args, class Scale = EpiNoScale>
__device__ __forceinline__ void gemm_epilogue_wide_dispatch(const Args& p, f32x4 (&acc)[MI][NT], int mbase, int nbase,
                                                            int lane, char* st, const Scale& sc = Scale()) {
  switch (p.act) {
    case ACT_QUICKGELU: gemm_epilogue_wide<ACT_QUICKGELU, MI, NT>(p, acc, mbase, nbase, lane, st, sc); break;
    case ACT_GELU_ERF: gemm_epilogue_wide<ACT_GELU_ERF, MI, NT>(p, acc, mbase, nbase, lane, st, sc); break;
    case ACT_SWIGLU:
      if constexpr (NT % 2 == 0) gemm_epilogue_wide<ACT_SWIGLU, MI, NT>(p, acc, mbase, nbase, lane, st, sc);
      break;
    default: gemm_epilogue_wide<ACT_NONE, MI, NT>(p, acc, mbase, nbase, lane, st, sc); break;
  }
}
