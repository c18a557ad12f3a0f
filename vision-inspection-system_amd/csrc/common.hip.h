// Device-side helpers shared by every gfx950 kernel in this library.
// Written for CDNA4 only: 64-lane wavefronts, MFMA 16x16x32 bf16, 160 KiB LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bf16 bits in memory
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define VIS_WAVE 64

// status codes returned by every extern "C" launcher
#define VIS_OK 0
#define VIS_ERR_ARG 1     // shape / alignment precondition violated
#define VIS_ERR_LAUNCH 2  // hipGetLastError() after the launch was not hipSuccess
#define VIS_ERR_UNSUPPORTED 3  // valid arguments, but this kernel form does not cover the shape on this device (vis_decode_chain:
                               // head shape outside the chained form, or a grid larger than the device holds resident); nothing
                               // was launched and the caller uses the equivalent separate launches

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float((uint32_t)v << 16); }

// plain cast: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN stays NaN)
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}

// one v_cvt_pk_bf16_f32 (RNE), lo in bits 0..15 (the scalar form costs two converts + shift + or)
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

__device__ __forceinline__ void unpack8(const u32x4& v, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(v[i] << 16);
    f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
  }
}

__device__ __forceinline__ u32x4 pack8(const float* f) {
  u32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = pack2bf(f[2 * i], f[2 * i + 1]);
  return v;
}

// Several requests of a prompt-pass group in ONE launch (vis_qkv_rope_split_many / vis_attn_prefill_pairs_many): blockIdx.z is the
// request; its tensors sit at uniform strides except the KV cache, whose slot is arbitrary (element offsets from the first pointer).
#define VIS_MAX_REQ 8
struct ReqOffsets {
  long long kv[VIS_MAX_REQ];   // k / v cache of request r = base + kv[r]
  long long qkv_bs, q_bs, vt_bs, o_bs;
  int work_bs;                 // work items between two requests' lists (0: one list for all)
};
// kv[z] by a chain of scalar selects: a dynamically indexed kernel-argument array is copied to scratch first
__device__ __forceinline__ long long req_kv(const ReqOffsets& r, int z) {
  long long o = r.kv[0];
#pragma unroll
  for (int i = 1; i < VIS_MAX_REQ; ++i) o = (z == i) ? r.kv[i] : o;
  return o;
}
static inline void req_offsets_none(ReqOffsets& r) {
  for (int i = 0; i < VIS_MAX_REQ; ++i) r.kv[i] = 0;
  r.qkv_bs = r.q_bs = r.vt_bs = r.o_bs = 0;
  r.work_bs = 0;
}

// One column of a finalised split-K projection row: the fixed-order sum `a` of its f32 partials, times the e4m3 scales
// (fp8 form: sxb * swn), plus the bias - the arithmetic of skinny_finalize_kernel's plain mode, shared with the decode attention
// kernels that finalise the qkv row themselves (vis_decode_attn_parts), so both produce the same bits.  Contraction is off:
// whether `a * s + bias` becomes an FMA must not depend on the kernel the expression is inlined into.
__device__ __forceinline__ float fin_plain_value(float a, bool scaled, float sxb, float swn, bool has_bias, float bias) {
#pragma clang fp contract(off)
  float v = a;
  if (scaled) v = v * (sxb * swn);
  if (has_bias) v = v + bias;
  return v;
}

// x * sigmoid(k x) = x / (1 + e^(-k x)) with ONE v_exp_f32 and ONE v_rcp_f32 (each 1 ulp): the IEEE division the
// plain expression compiles to costs ~10 VALU instructions per element, which made the activation epilogues of the
// prefill GEMMs (128 elements per thread on a 256 x 256 tile) 16 us per tile round (tools/gemm_kscan.py: fixed cost
// 32.4 us per round with QuickGELU against 16.4 us without).  SiLU: k = 1; QuickGELU: k = 1.702.
#define VIS_LOG2E 1.4426950408889634f
__device__ __forceinline__ float x_sigmoid(float x, float k_log2e) {
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-k_log2e * x));
}
__device__ __forceinline__ float silu_fast(float x) { return x_sigmoid(x, VIS_LOG2E); }
__device__ __forceinline__ float quickgelu_fast(float x) { return x_sigmoid(x, 1.702f * VIS_LOG2E); }

// Cross-row exchanges without the LDS crossbar: gfx950's v_permlane16_swap / v_permlane32_swap swap 16- / 32-lane
// halves between two registers in one VALU op (a ds_bpermute round trip costs ~100+ cycles of latency).
// With both operands a copy of x: swap16 -> {rows 0,0,2,2 | rows 1,1,3,3}, swap32 -> {low half twice | high half twice}.
__device__ __forceinline__ float hsum4(float x) {  // sum over lanes l, l^16, l^32, l^48
  const uint32_t u = __float_as_uint(x);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const uint32_t v = __float_as_uint(s);
  const auto b = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

__device__ __forceinline__ float hmax4(float x) {  // max over lanes l, l^16, l^32, l^48
  const uint32_t u = __float_as_uint(x);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const float s = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  const uint32_t v = __float_as_uint(s);
  const auto b = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// Intra-row (16 lanes) butterflies on the DPP path: quad_perm xor-1, xor-2, then row_half_mirror and row_mirror
// (valid because the partial results are already symmetric) - four v_add_f32_dpp / v_max_f32_dpp instead of four
// ds_bpermute round trips through the LDS crossbar (__shfl_xor).  The two cross-row steps are hsum4 / hmax4.
#define VIS_DPP(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xf, 0xf, true))

__device__ __forceinline__ float wave_sum(float v) {
  v += VIS_DPP(v, 0xB1);   // quad_perm [1,0,3,2]
  v += VIS_DPP(v, 0x4E);   // quad_perm [2,3,0,1]
  v += VIS_DPP(v, 0x141);  // row_half_mirror
  v += VIS_DPP(v, 0x140);  // row_mirror
  return hsum4(v);
}

__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, VIS_DPP(v, 0xB1));
  v = fmaxf(v, VIS_DPP(v, 0x4E));
  v = fmaxf(v, VIS_DPP(v, 0x141));
  v = fmaxf(v, VIS_DPP(v, 0x140));
  return hmax4(v);
}

// Bijective XCD-aware remap of a linear workgroup id: blocks that share
// (id % 8) share an XCD/L2, so give each XCD a contiguous chunk of tiles.
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (orig >> 3);
}

// hipGetLastError() is sticky per thread and the host application (PyTorch) leaves benign codes such as
// hipErrorNotReady behind, so every launcher clears the slot before launching and reads it after.
static inline void vis_clear_error() { (void)hipGetLastError(); }
static inline int vis_check_launch() {
  return hipGetLastError() == hipSuccess ? VIS_OK : VIS_ERR_LAUNCH;
}
