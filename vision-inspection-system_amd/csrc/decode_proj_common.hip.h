// Shared by the two forms of the batched decode projection (decode_stream.hip: stream-K ranges with a last-arriver
// reduction; decode_colpar.hip: one whole-K column slab per workgroup): arguments, MX helpers, the deferred RMSNorm's row
// factors and the epilogue of one wave's 32-column slab.
#pragma once
#include "common.hip.h"

#define DS_BN 128                      // tile columns
#define DS_ROWB 128                    // bytes per tile row and K-step (64 bf16 / 128 e4m3)
#define DS_MAX_WG 256                  // one per CU
#define DS_MAX_SEGS 16                 // segments a tile may be cut into
#define DS_SSQ_LD 64                   // row stride of the ssq tile partials
#define DS_CNT_BYTES 16384             // arrival counters: the first 16 KB of every workspace, whatever the shape - one
                                       // workspace serves all projections of a step (N <= 4096 x 128 columns)

enum { DS_PLAIN = 0, DS_SWIGLU = 1, DS_RESID_NORMW = 2 };

struct DsArgs {
  const char* A;         // [M][lda] bf16, or e4m3 bytes (FP8)
  const char* W;         // [N][ldw] bf16, or e4m3 bytes (FP8)
  const uint8_t* As;     // FP8: E8M0 block scales of A, [M][ldas], one per 32 columns
  const float* sw;       // FP8: per-output-row weight scales [N]
  float* part;           // segment blocks of split tiles
  int* cnt;              // [tiles] arrival counters, zero between launches
  void* C;               // main output (PLAIN: bf16 or f32 [M][ldc]; SWIGLU: bf16 [M][ldc] of N/2 columns; RESID: y) or null
  bf16_t* Cw;            // RESID_NORMW: bf16(y * nw[n]) or null
  uint8_t* Cq;           // MX copy of the row the next projection consumes (RESID: y * nw, SWIGLU: act) or null
  uint8_t* Cqs;          // its E8M0 scales [M][ldcqs]
  const bf16_t* bias;    // PLAIN: [N] or null
  const bf16_t* R;       // RESID_NORMW: [M][ldr]
  const bf16_t* nw;      // RESID_NORMW: [N]
  const float* ssq_in;   // [tiles_in][DS_SSQ_LD] partial sums of squares of the row A was derived from, or null (rs = 1)
  float* ssq_out;        // RESID_NORMW: [tiles][DS_SSQ_LD]
  int M, N, lda, ldw, ldas, ldc, ldr, ldcq, ldcqs;
  int nk_all, total, spb, lcm;
  int mode, out_f32, tiles_in;
  float inv_norm_dim, eps;
};

// global index of the segment that starts at step s0 of the (tile, K-step) sequence: the sequence is cut at every multiple
// of nk (tile seams) and of spb (workgroup seams)
__device__ __forceinline__ int ds_seg_id(int s0, int nk, int spb, int lcm) { return s0 / nk + s0 / spb - s0 / lcm; }

// ---- 16-byte agent-coherent (sc1) accesses to the segment blocks: buffer instructions with the cache-policy operand
__device__ __forceinline__ void ds_store_sc1(const __amdgpu_buffer_rsrc_t rs, unsigned off, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, off, 0, 16 /* sc1 */);
}
__device__ __forceinline__ f32x4 ds_load_sc1(const __amdgpu_buffer_rsrc_t rs, unsigned off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16 /* sc1 */));
}

// E8M0 byte of the smallest power of two X with amax / X <= 448 (the e4m3 maximum); amax == 0 -> 2^-127 (all codes 0)
__device__ __forceinline__ int ds_mx_scale_byte(float amax) {
  const uint32_t u = __float_as_uint(amax);
  const int E = (int)((u >> 23) & 0xff);                 // biased exponent of amax (0 for zero / denormal)
  const uint32_t man = u & 0x7fffffu;
  int b = E - ((man <= 0x600000u) ? 8 : 7);              // mantissa <= 1.75 -> amax / 2^(E - 8) <= 448
  return b < 0 ? 0 : (b > 254 ? 254 : b);
}
__device__ __forceinline__ float ds_mx_inv_scale(int byte) {   // 2^(127 - byte), exact
  return __uint_as_float((uint32_t)(254 - byte) << 23);
}


// rs[row] = rsqrt(sum over the K / 32 unit partials of the row's squares / norm_dim + eps) for the 64 rows of a launch, into
// rs_s[64]; all 256 threads of the workgroup call it (two barriers).  Wave w sums units 32 w .. 32 w + 31 (every partial
// requested at once: coalesced 256-byte rows of ssq_in[unit][64]), the four wave sums are added in wave order: a fixed order,
// the same for every batch size.  `sq` holds the caller's early loads (ds_row_factor_loads) so that the round trip can hide
// under other memory traffic.
struct DsRowLoads { float v[32]; };
__device__ __forceinline__ void ds_row_factor_loads(const DsArgs& p, DsRowLoads& sq, int wn, int lane) {
  if (!p.ssq_in) return;
#pragma unroll
  for (int t = 0; t < 32; ++t) sq.v[t] = p.ssq_in[(size_t)min(32 * wn + t, p.tiles_in - 1) * DS_SSQ_LD + lane];
}
__device__ __forceinline__ void ds_row_factors(const DsArgs& p, const DsRowLoads& sq, float (*part_s)[64], float* rs_s, int wn, int lane) {
  if (p.ssq_in) {
    float tot = 0.f;
#pragma unroll
    for (int t = 0; t < 32; ++t)
      if (32 * wn + t < p.tiles_in) tot += sq.v[t];
    part_s[wn][lane] = tot;
  }
  __syncthreads();
  if (wn == 0) {
    float tot = 0.f;
    if (p.ssq_in) {
      tot = ((part_s[0][lane] + part_s[1][lane]) + part_s[2][lane]) + part_s[3][lane];
      for (int t = 128; t < p.tiles_in; ++t) tot += p.ssq_in[(size_t)t * DS_SSQ_LD + lane];   // (norm dims > 4096: none today)
    }
    rs_s[lane] = p.ssq_in ? rsqrtf(tot * p.inv_norm_dim + p.eps) : 1.0f;
  }
  __syncthreads();
}

// Epilogue of ONE WAVE's slab: 32 consecutive columns starting at col0 (a multiple of 32), the 16 MBW rows from row block
// mb0 on; the lane holds D[n = col0 + 16 j + 4 h + r][m = 16 (mb0 + mb) + l15], r = 0..3 (the MFMA's C layout with the
// weights as the A operand).  rs: the rows' deferred-norm factors.  SWIGLU with an MX output pairs this wave with its
// neighbour (a block of 32 act columns = two slabs): pair_s = LDS [4][64] floats, slot = this wave's index there, every
// wave of the workgroup is inside this call at the same time (two barriers) - the stream-K form only.
template <bool FP8, int MBW>
__device__ __forceinline__ void ds_epilogue(const DsArgs& p, f32x4 (&acc)[MBW][2], int mb0, int col0, const float (&rs)[MBW],
                                            int lane, float (*pair_s)[64], int slot) {
  const int l15 = lane & 15, h = lane >> 4;
  const int nb = col0 + 4 * h;   // column of (j = 0, r = 0)
  if (p.mode == DS_SWIGLU) {
    // j = 0: 16 gate columns, j = 1: the matching up columns -> act column col0 / 2 + 4 h + r
    const int o = (col0 >> 1) + 4 * h;
    const bool live_n = nb + 16 < p.N;
    f32x4 sg = (f32x4){1.f, 1.f, 1.f, 1.f}, su = sg;
    if (FP8 && live_n) { sg = *(const f32x4*)(p.sw + nb); su = *(const f32x4*)(p.sw + nb + 16); }
    float act[MBW][4];
    float amax[MBW];
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb) {
      amax[mb] = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float gq = acc[mb][0][r] * (FP8 ? sg[r] * rs[mb] : rs[mb]), uq = acc[mb][1][r] * (FP8 ? su[r] * rs[mb] : rs[mb]);
        // the bf16 value a bf16 consumer reads (the MX copy quantises that value, so both outputs agree)
        act[mb][r] = bf2f(f2bf(silu_fast(gq) * uq));
        amax[mb] = fmaxf(amax[mb], fabsf(act[mb][r]));
      }
    }
    if (p.C) {
#pragma unroll
      for (int mb = 0; mb < MBW; ++mb) {
        const int m = (mb0 + mb) * 16 + l15;
        if (m < p.M && live_n) {
          u32x2 q;
          q[0] = pack2bf(act[mb][0], act[mb][1]);
          q[1] = pack2bf(act[mb][2], act[mb][3]);
          *(u32x2*)((bf16_t*)p.C + (size_t)m * p.ldc + o) = q;
        }
      }
    }
    if (p.Cq) {   // MX block = 32 act columns = this wave's 16 and its neighbour's (slot ^ 1)
#pragma unroll
      for (int mb = 0; mb < MBW; ++mb) {
        amax[mb] = hmax4(live_n ? amax[mb] : 0.f);
        if (h == 0) pair_s[slot][mb * 16 + l15] = amax[mb];
      }
      __syncthreads();
#pragma unroll
      for (int mb = 0; mb < MBW; ++mb) {
        const int m = (mb0 + mb) * 16 + l15;
        const int sb = ds_mx_scale_byte(fmaxf(pair_s[slot][mb * 16 + l15], pair_s[slot ^ 1][mb * 16 + l15]));
        const float inv = ds_mx_inv_scale(sb);
        if (m < p.M && live_n) {
          int w = 0;
          w = __builtin_amdgcn_cvt_pk_fp8_f32(act[mb][0] * inv, act[mb][1] * inv, w, false);
          w = __builtin_amdgcn_cvt_pk_fp8_f32(act[mb][2] * inv, act[mb][3] * inv, w, true);
          *(uint32_t*)(p.Cq + (size_t)m * p.ldcq + o) = (uint32_t)w;
          if (h == 0 && (slot & 1) == 0) p.Cqs[(size_t)m * p.ldcqs + (o >> 5)] = (uint8_t)sb;
        }
      }
      __syncthreads();   // pair_s is reused by the next tile's epilogue
    }
    return;
  }
  // ---- PLAIN / RESID_NORMW: 8 values per (lane, mb): columns nb + 16 j + r
  f32x4 swv[2] = {(f32x4){1.f, 1.f, 1.f, 1.f}, (f32x4){1.f, 1.f, 1.f, 1.f}};
  u32x2 bb[2] = {(u32x2){0u, 0u}, (u32x2){0u, 0u}}, nwv[2] = {(u32x2){0u, 0u}, (u32x2){0u, 0u}};
  bool live[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = nb + 16 * j;
    live[j] = n < p.N;
    const int nc = live[j] ? n : 0;
    if (FP8) swv[j] = *(const f32x4*)(p.sw + nc);
    if (p.bias) bb[j] = *(const u32x2*)(p.bias + nc);
    if (p.mode == DS_RESID_NORMW) nwv[j] = *(const u32x2*)(p.nw + nc);
  }
#pragma unroll
  for (int mb = 0; mb < MBW; ++mb) {
    const int m = (mb0 + mb) * 16 + l15;
    const bool row_ok = m < p.M;
    float ssq = 0.f, amax = 0.f;
    float u8[2][4];   // the value the next projection consumes (RESID: y * nw)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = nb + 16 * j;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[mb][j][r] * (FP8 ? swv[j][r] * rs[mb] : rs[mb]);
      if (p.mode == DS_PLAIN) {
        if (p.bias) {
          v[0] += __uint_as_float(bb[j][0] << 16); v[1] += __uint_as_float(bb[j][0] & 0xffff0000u);
          v[2] += __uint_as_float(bb[j][1] << 16); v[3] += __uint_as_float(bb[j][1] & 0xffff0000u);
        }
        if (row_ok && live[j]) {
          if (p.out_f32) {
            *(f32x4*)((float*)p.C + (size_t)m * p.ldc + n) = (f32x4){v[0], v[1], v[2], v[3]};
          } else {
            u32x2 q;
            q[0] = pack2bf(v[0], v[1]);
            q[1] = pack2bf(v[2], v[3]);
            *(u32x2*)((bf16_t*)p.C + (size_t)m * p.ldc + n) = q;
          }
        }
        continue;
      }
      // RESID_NORMW: y = bf16(v + R); yw = bf16(y * nw)
      u32x2 rr = (u32x2){0u, 0u};
      if (row_ok && live[j]) rr = *(const u32x2*)(p.R + (size_t)m * p.ldr + n);
      v[0] += __uint_as_float(rr[0] << 16); v[1] += __uint_as_float(rr[0] & 0xffff0000u);
      v[2] += __uint_as_float(rr[1] << 16); v[3] += __uint_as_float(rr[1] & 0xffff0000u);
      u32x2 q;
      q[0] = pack2bf(v[0], v[1]);
      q[1] = pack2bf(v[2], v[3]);
      const float y[4] = {__uint_as_float(q[0] << 16), __uint_as_float(q[0] & 0xffff0000u), __uint_as_float(q[1] << 16),
                          __uint_as_float(q[1] & 0xffff0000u)};
      const float g[4] = {__uint_as_float(nwv[j][0] << 16), __uint_as_float(nwv[j][0] & 0xffff0000u),
                          __uint_as_float(nwv[j][1] << 16), __uint_as_float(nwv[j][1] & 0xffff0000u)};
      u32x2 qw;
      qw[0] = pack2bf(y[0] * g[0], y[1] * g[1]);
      qw[1] = pack2bf(y[2] * g[2], y[3] * g[3]);
      u8[j][0] = __uint_as_float(qw[0] << 16); u8[j][1] = __uint_as_float(qw[0] & 0xffff0000u);
      u8[j][2] = __uint_as_float(qw[1] << 16); u8[j][3] = __uint_as_float(qw[1] & 0xffff0000u);
      if (row_ok && live[j]) {
        if (p.C) *(u32x2*)((bf16_t*)p.C + (size_t)m * p.ldc + n) = q;
        if (p.Cw) *(u32x2*)(p.Cw + (size_t)m * p.ldc + n) = qw;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          ssq += y[r] * y[r];
          amax = fmaxf(amax, fabsf(u8[j][r]));
        }
      }
    }
    if (p.mode != DS_RESID_NORMW) continue;
    // the slab's sum of squares of row m: 8 lane-local values, then the four lanes that share the row (fixed order)
    const float s4 = hsum4(ssq);
    if (h == 0 && row_ok && live[0]) p.ssq_out[(size_t)(col0 >> 5) * DS_SSQ_LD + m] = s4;
    if (p.Cq) {   // MX block = this slab's 32 columns of row m
      const int sb = ds_mx_scale_byte(hmax4(amax));
      const float inv = ds_mx_inv_scale(sb);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (!(row_ok && live[j])) continue;
        int w = 0;
        w = __builtin_amdgcn_cvt_pk_fp8_f32(u8[j][0] * inv, u8[j][1] * inv, w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(u8[j][2] * inv, u8[j][3] * inv, w, true);
        *(uint32_t*)(p.Cq + (size_t)m * p.ldcq + nb + 16 * j) = (uint32_t)w;
      }
      if (h == 0 && row_ok && live[0]) p.Cqs[(size_t)m * p.ldcqs + (col0 >> 5)] = (uint8_t)sb;
    }
  }
}

// argument checks shared by the launchers of both forms
static inline int ds_check_common(const DsArgs& p, int B, int K, const void* ws, int mode, bool fp8) {
  if (!p.A || !p.W || !ws || B <= 0 || B > 64 || p.N <= 0 || K <= 0) return VIS_ERR_ARG;
  if (K % (fp8 ? 128 : 64) != 0 || p.N % 4 != 0) return VIS_ERR_ARG;
  if (((uintptr_t)p.A | (uintptr_t)p.W | (uintptr_t)ws) & 15) return VIS_ERR_ARG;
  if (mode != DS_PLAIN && mode != DS_SWIGLU && mode != DS_RESID_NORMW) return VIS_ERR_ARG;
  if (mode == DS_PLAIN && (!p.C || p.R || p.nw || p.Cw || p.Cq || p.ssq_out || p.ldc % 4 != 0)) return VIS_ERR_ARG;
  if (mode == DS_SWIGLU && (p.N % 64 != 0 || p.bias || p.R || p.nw || p.Cw || p.ssq_out || p.out_f32 || (!p.C && !p.Cq) || p.ldc % 4 != 0))
    return VIS_ERR_ARG;
  if (mode == DS_RESID_NORMW && (!p.R || !p.nw || !p.ssq_out || p.bias || p.out_f32 || (!p.C && !p.Cw && !p.Cq) ||
                                 p.ldc % 4 != 0 || p.ldr % 4 != 0 || p.N % 32 != 0))
    return VIS_ERR_ARG;
  if ((p.Cq != nullptr) != (p.Cqs != nullptr) || (p.Cq && (p.ldcq % 4 != 0 || ((uintptr_t)p.Cq & 3)))) return VIS_ERR_ARG;
  if (p.Cq && mode == DS_PLAIN) return VIS_ERR_ARG;
  if (((uintptr_t)p.C | (uintptr_t)p.Cw | (uintptr_t)p.bias | (uintptr_t)p.R | (uintptr_t)p.nw) & 7) return VIS_ERR_ARG;
  if (p.out_f32 && ((uintptr_t)p.C & 15)) return VIS_ERR_ARG;
  if (p.ssq_in && (p.tiles_in <= 0 || p.tiles_in > 128)) return VIS_ERR_ARG;
  if (((uintptr_t)p.ssq_in | (uintptr_t)p.ssq_out) & 3) return VIS_ERR_ARG;
  return VIS_OK;
}
