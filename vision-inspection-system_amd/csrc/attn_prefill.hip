// K6 + K7: flash-style prefill attention for gfx950 (MI355X), one kernel for
//  * the ViT: non-causal, varlen segments (cu_seqlens), head_dim 80
//    (TF:models/qwen2_vl/modeling_qwen2_vl.py:356-423; Qwen2.5-VL windows are
//    just shorter segments), and
//  * the LLM: causal GQA prefill, head_dim 128 (TF:...:508-556, :318-340).
//
// Work decomposition: the host passes a work list of {q0, qn<=128, k0, k1}
// tiles (one workgroup each; heads on grid.x, items on grid.y): query rows
// [q0,q0+qn) attend keys [k0,k1) (and key <= query when causal).  A work item
// never straddles a segment, so cu_seqlens, windows and plain causal prefill are
// all the same kernel.  Workgroups are dispatched heads-fastest, i.e. in work-list
// order: the host lists full 128-row items first and may cut the remainder into
// cheaper 64-row items that fill the last, partial round of the 512 workgroup
// slots (ViT, 4900 patches x 16 heads: 624 full items = 2 rounds at 61 %; 512 full
// + 208 half items = ~1.55 rounds).  With 16 heads and round-robin XCD dispatch an
// XCD only ever sees two heads' K/V, which fit its L2.
//
// Per workgroup: 256 threads = 4 waves; a wave owns 32 query rows (two 16-row
// blocks) of a 65..128-row item, or 16 rows (one block) of a <= 64-row item, and
// walks KV tiles of 64 keys that the whole workgroup stages through LDS
// (double-buffered, one barrier per tile): head_dim 128 stages through registers (2 workgroups per CU);
// head_dim 80 stages by LDS-DMA with the swizzles on the source address, recomputes its source offsets per tile
// and runs the 16 leftover dims of QK^T on v_mfma_f32_16x16x16_bf16 - together that fits 168 VGPRs, i.e. THREE
// workgroups per CU (PMC: the 2-per-CU kernel left both the MFMA and the VALU pipe < 40 % busy, waves waiting).
//
// MFMA formulation (v_mfma_f32_16x16x32_bf16, f32 accumulate, f32 softmax):
//  * scores are computed TRANSPOSED,  S^T[key][q] = K * Q^T  (A = K rows from
//    LDS via ds_read_b128, B = Q^T held in registers for the whole kernel), so
//    the accumulator has the query on the lane (col) and 4 keys per register
//    group: the row max/sum is 16 in-lane ops + two xor-shuffles (16, 32).
//  * the exponentiated tile is already the A operand of P*V: two S^T blocks
//    (32 keys) pack in-lane into one bf16x8 fragment; the k-slot order this
//    implies (slot (h,j) <-> key 16*(j>>2) + 4h + (j&3)) is applied to the B
//    operand instead: vis_qkv_rope_split writes V^T ([d][key]) with exactly that
//    column order inside every 32-key group, so a lane's B fragment is one
//    aligned 16-byte chunk (one ds_read_b128; it used to be two ds_read_b64 plus
//    register moves).  No cross-lane movement of P and no transposed LDS read.
//  * K tile: 256-B rows, 16-B chunk index XOR (row & 15)   -> conflict-free b128
//    (head_dim 80: 224-B padded rows - 16-B slot (14 r + c) mod 16 is distinct over every ds_read_b128
//    lane group - with zero-filled d 80..95);
//    V^T tile: 128-B rows, 16-B chunk index XOR ((d >> 1) & 7) -> conflict-free b128.
// Online softmax in the log2 domain with a finite -1e30 sentinel (no inf-inf).
// Output O[s][head*HD + d] is staged through LDS and written as whole 16-B
// chunks so the following projection GEMM reads a plain row-major matrix.
#include "common.hip.h"
#include <type_traits>

struct AttnArgs {
  const bf16_t* Q;   // [Hq][Sq][HD]
  const bf16_t* K;   // [Hkv][k_tokens][HD]
  const bf16_t* Vt;  // [Hkv][HD][vt_ld]
  bf16_t* O;         // [Sq][ldo]
  const int4* work;  // [grid.x] {q0, qn, k0, k1}
  int Sq, k_tokens, vt_ld, ldo, group;
  float scale_log2;
  int q_row0;        // work-list query rows are positions of the whole sequence; Q / O hold rows q_row0 .. q_row0 + Sq - 1
  // key-split items of attn_vit32_kernel (vis_attn_prefill_split): counters [n_pairs * Hq] then partial blocks
  int* ws_count;
  float* ws_part;
  int n_pairs;
  // attn_prefill_pair_kernel only (vis_attn_prefill_pairs_many): request blockIdx.z reads Q + z * q_bs, K + kv[z], Vt + z * vt_bs
  // and writes O + z * o_bs
  ReqOffsets req;
#ifdef VIT_VARIANTS_STAMPS
  unsigned long long* stamps;   // tools/probes/attn_vit_variants.hip only: s_memtime stamps of workgroup (0, 0)
#endif
};

#define ATT_NEG (-1.0e30f)

// Timing probes (tools/probes/attn_probe.sh builds this file with -DATT_PROBE=<bits>; the product build has 0 and
// every probe branch is compiled out).  Results of a probe build are WRONG by construction - only its duration is read.
//   1: stage only the first two key tiles (no K/V traffic afterwards)     2: v_exp_f32 -> v_mul_f32 (no transcendental)
//   4: skip the P*V MFMAs                                                 8: skip the QK^T MFMAs
//  16: no barrier / no wait at the end of a tile
#ifndef ATT_PROBE
#define ATT_PROBE 0
#endif
__device__ __forceinline__ float att_exp2(float x) {
  if constexpr ((ATT_PROBE & 2) != 0) return x * 0.00390625f;
  else return __builtin_amdgcn_exp2f(x);
}

__device__ __forceinline__ float att_max3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float att_max(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// The online softmax keeps a REFERENCE m per query instead of the strict running maximum: m is the first tile's maximum
// and afterwards follows a tile's maximum only when that exceeds it by more than ATT_LAZY (log2 units), so
// p = exp2(s - m) <= 2^ATT_LAZY = 256 instead of <= 1.  Still exact after the final division (numerator and denominator
// carry the same m; bf16 keeps its relative precision, the f32 sums stay far from overflow), and after the first tile of
// a row nothing moves on almost every tile: no alpha, no O rescale (48 multiplies per tile in the 32-wide kernel, taken
// on ~2/3 of the tiles with the strict maximum on N(0,1) scores), no per-score subtraction (the kernels keep -m as the
// C operand of the first QK^T MFMA).  The decision is per lane = per query, a function of that query's own keys in the
// fixed absolute tile order, so results still do not depend on what shares the wave, the launch or the batch.
// -DATT_LAZY=0.0f gives the strict maximum.
#ifndef ATT_LAZY
#define ATT_LAZY 8.0f
#endif
// (Packing the score scaling as v_pk_fma_f32 - two scores per instruction - was measured SLOWER than two v_fma_f32:
// 189-191 us against 184-185 us on the ViT shape, same box, tools/probes/attn_ab.sh; a packed f32 op holds the issue port
// for both halves.)
// max over lanes l, l^16, l^32, l^48 (hmax4 of common.hip.h without the canonicalising self-maxes)
__device__ __forceinline__ float att_hmax4(float x) {
  const uint32_t u = __float_as_uint(x);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const float s = att_max(__uint_as_float(a[0]), __uint_as_float(a[1]));
  const uint32_t v = __float_as_uint(s);
  const auto b = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  return att_max(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// Q fragments are multiplied by scale_log2 once, when they are loaded (one more bf16 rounding of Q, no multiply per score
// afterwards): the QK^T MFMAs then produce scores in the log2 domain, and with the running reference -m as the C operand
// of their first k-step the accumulators hold s - m directly.
__device__ __forceinline__ bf16x8 att_scaled_q8(const bf16_t* q, float scale) {
  float f[8];
  unpack8(*(const u32x4*)q, f);
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] *= scale;
  return __builtin_bit_cast(bf16x8, pack8(f));
}

template <int HD, bool CAUSAL>
// (head_dim 80 here is the A/B fallback of attn_vit32_kernel and the causal form nothing in the engines uses: it is
// compiled for two workgroups per CU since the -m seed registers arrived - at three it spilled ten dwords.)
__global__ __launch_bounds__(256, 2) void attn_prefill_kernel(AttnArgs p) {
  constexpr int DKS = (HD + 31) / 32;           // QK^T k-steps over d: 4 / 3
  constexpr int ND = HD / 16;                   // P*V output blocks over d: 8 / 5
  constexpr int KCH = HD / 8;                   // valid 16-B chunks per K row: 16 / 10
  // head_dim 80: K / V^T tiles arrive by LDS-DMA (global_load_lds, lane-linear LDS image, the swizzle is applied to the
  // SOURCE address) - no staging VGPRs, which is what lets three workgroups share a CU (<= 168 VGPRs).
  constexpr bool DMA = (HD == 80);
  constexpr int KROW = (HD == 128) ? 256 : 160;         // LDS bytes per K row (d = 80: dense rows)
  constexpr int K_BYTES = DMA ? 12288 : 64 * KROW;      // DMA: 768 16-byte slots (3 per thread), 640 used
  constexpr int V_BYTES = DMA ? 12288 : HD * 128;
  constexpr int BUF = K_BYTES + V_BYTES;
  constexpr int K_ITERS = (64 * KCH + 255) / 256;  // 4 / 3
  constexpr int V_ITERS = (HD * 8 + 255) / 256;    // 4 / 3
  constexpr int OROW = HD * 2 + 16;             // staging row bytes for the O tile
  static_assert(2 * BUF >= 4 * 32 * OROW, "O staging must fit in the KV buffers");
  __shared__ __attribute__((aligned(16))) char lds[2 * BUF];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, h = lane >> 4;
  const int head = blockIdx.x, hkv = head / p.group;
  // request blockIdx.z of vis_attn_prefill_rows_many (locals, not updates of the by-value argument struct: see the pair kernel)
  const int z = blockIdx.z;
  const bf16_t* const r_Q = p.Q + z * p.req.q_bs;
  const bf16_t* const r_K = p.K + req_kv(p.req, z);
  const bf16_t* const r_Vt = p.Vt + z * p.req.vt_bs;
  bf16_t* const r_O = p.O + z * p.req.o_bs;
  const int4 wk = p.work[blockIdx.y + z * p.req.work_bs];
  const int q0 = wk.x, qn = wk.y, k0 = wk.z;
  const int k1 = CAUSAL ? min(wk.w, q0 + qn) : wk.w;
  const int kt_begin = k0 & ~63;
  const int nt = (k1 - kt_begin + 63) >> 6;
  // 16-row query blocks per wave (workgroup-uniform); the one-block form exists for head_dim 80 only (the
  // head_dim 128 kernel is at the register limit, and its causal grid already fits one round)
  const int nqb = (HD == 80 && qn <= 64) ? 1 : 2;
  const int wrows = 16 * nqb;
  const int wq0 = q0 + wave * wrows;        // first query row of this wave

  const bf16_t* Kh = r_K + (size_t)hkv * p.k_tokens * HD;
  const bf16_t* Vh = r_Vt + (size_t)hkv * HD * p.vt_ld;

  // ---- Q^T fragments (B operand), kept in registers
  // head_dim 80 = two 32-wide k-steps + one 16-wide tail on v_mfma_f32_16x16x16_bf16 (lane: row l&15, k = 4(l>>4)+j):
  // no zero-padded third 32-step (17 % fewer QK^T MFMA cycles, 4 fewer VGPRs, half the K-fragment bytes for it)
  constexpr int DKF = HD / 32;                 // full 32-wide k-steps: 4 / 2
  constexpr bool TAIL16 = (HD % 32) == 16;
  typedef short bf16x4s __attribute__((ext_vector_type(4)));
  bf16x8 qf[2][DKF];
  bf16x4s qt[2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int qrow = min(wq0 + qb * 16 + l15, p.q_row0 + p.Sq - 1) - p.q_row0;
    const bf16_t* qp = r_Q + ((size_t)head * p.Sq + qrow) * HD;
#pragma unroll
    for (int ds = 0; ds < DKF; ++ds) qf[qb][ds] = att_scaled_q8(qp + ds * 32 + 8 * h, p.scale_log2);
    if constexpr (TAIL16) {
      const u32x2 raw = *(const u32x2*)(qp + DKF * 32 + 4 * h);
      u32x2 sc;
      sc[0] = pack2bf(__uint_as_float(raw[0] << 16) * p.scale_log2, __uint_as_float(raw[0] & 0xffff0000u) * p.scale_log2);
      sc[1] = pack2bf(__uint_as_float(raw[1] << 16) * p.scale_log2, __uint_as_float(raw[1] & 0xffff0000u) * p.scale_log2);
      qt[qb] = __builtin_bit_cast(bf16x4s, sc);
    } else {
      qt[qb] = (bf16x4s){0, 0, 0, 0};
    }
  }


  // per-thread staging slots, computed once (no div/mod inside the KV loop)
  int k_goff[K_ITERS], k_loff[K_ITERS], k_row[K_ITERS], v_goff[V_ITERS], v_loff[V_ITERS];
  u32x4 kreg[DMA ? 1 : K_ITERS], vreg[DMA ? 1 : V_ITERS];
  if constexpr (DMA) {
    // LDS slot p = i * 256 + tid (16 bytes each, lane-linear per wave instruction).  K slot p = (row p / 10, chunk c):
    // chunks 0..7 (the ds_read_b128 fragments) are stored in place - with 160-byte rows the b128 lane groups
    // {0-3, 12-15, 20-27} ... already hit 16 distinct 16-byte bank slots (rows 0-3 / 12-15 take the even slots with
    // chunk c0, rows 4-11 the odd ones with c0 + 1); r02 PMC showed the former XOR on these chunks CAUSED 2-way
    // conflicts (SQ_LDS_BANK_CONFLICT 1.6 x SQ_ACTIVE_INST_LDS).  Chunks 8, 9 (the 8-byte tail reads, two 32-lane
    // groups) keep c ^ ((row >> 3) & 1): rows r and r + 8 then land on different bank quads (dense 160-byte rows
    // alone would collide two-way); V^T slot p holds chunk c ^ ((d >> 1) & 7) of row d = p / 8, as before.
    // The per-lane source offsets (dk_off / dv_off below) are loop constants; they fit the 168-VGPR budget since the
    // denominators left the MFMA pipe (MFMA_SUM == false for this head_dim).
#pragma unroll
    for (int i = 0; i < K_ITERS; ++i) { k_row[i] = 0; k_goff[i] = 0; k_loff[i] = 0; }
#pragma unroll
    for (int i = 0; i < V_ITERS; ++i) { v_goff[i] = 0; v_loff[i] = 0; }
  } else {
#pragma unroll
    for (int i = 0; i < K_ITERS; ++i) {
      const int it = min(tid + i * 256, 64 * KCH - 1);
      const int row = it / KCH, c = it - row * KCH;
      k_row[i] = row;
      k_goff[i] = c * 8;
      k_loff[i] = row * KROW + ((c ^ (row & 15)) << 4);
    }
#pragma unroll
    for (int i = 0; i < V_ITERS; ++i) {
      const int it = min(tid + i * 256, HD * 8 - 1);
      const int d = it >> 3, c = it & 7;
      v_goff[i] = d * p.vt_ld + c * 8;
      v_loff[i] = K_BYTES + d * 128 + ((c ^ ((d >> 1) & 7)) << 4);
    }
  }
  const int dma_base = __builtin_amdgcn_readfirstlane(tid >> 6) * 1024;
  uint32_t dk_off[3], dv_off[3];
  if constexpr (DMA) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int ps = i * 256 + tid;
      const int r0 = (ps * 6554) >> 16;                 // ps / 10 for ps < 768
      const int c = ps - r0 * 10, row = min(r0, 63);
      dk_off[i] = (uint32_t)row * (HD * 2) + ((c < 8 ? c : (c ^ ((row >> 3) & 1))) << 4);
      const int d = min(ps >> 3, HD - 1), cv = ps & 7;
      dv_off[i] = (uint32_t)d * (uint32_t)(p.vt_ld * 2) + ((cv ^ ((d >> 1) & 7)) << 4);
    }
  }
  // global -> LDS of one 64-key tile (DMA), or global -> registers (the head_dim 128 path stores them later)
  auto load_tile = [&](int kt, int buf) {
    if constexpr (DMA) {
      char* base = lds + buf * BUF + dma_base;
      // per-lane byte offsets inside a tile are loop constants (dk_off / dv_off, 6 VGPRs); only the wave-uniform tile
      // base moves.  The ragged last tile of a head (rows past k_tokens) recomputes clamped K addresses instead.
      const char* vbase = (const char*)(Vh + kt);
      if (kt + 64 <= p.k_tokens) {
        const char* kbase = (const char*)Kh + (size_t)kt * (HD * 2);
#pragma unroll
        for (int i = 0; i < 3; ++i)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kbase + dk_off[i]),
                                           (__attribute__((address_space(3))) void*)(base + i * 4096), 16, 0, 0);
      } else {
        const char* kbase = (const char*)Kh;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const int ps = i * 256 + tid;
          const int r0 = (ps * 6554) >> 16;                 // ps / 10 for ps < 768
          const int c = ps - r0 * 10, row = min(r0, 63);
          const int key = min(kt + row, p.k_tokens - 1);
          const uint32_t off = (uint32_t)key * (HD * 2) + ((c < 8 ? c : (c ^ ((row >> 3) & 1))) << 4);
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kbase + off),
                                           (__attribute__((address_space(3))) void*)(base + i * 4096), 16, 0, 0);
        }
      }
#pragma unroll
      for (int i = 0; i < 3; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vbase + dv_off[i]),
                                         (__attribute__((address_space(3))) void*)(base + K_BYTES + i * 4096), 16, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < K_ITERS; ++i) {
        const int key = min(kt + k_row[i], p.k_tokens - 1);
        kreg[i] = *(const u32x4*)(Kh + (size_t)key * HD + k_goff[i]);
      }
#pragma unroll
      for (int i = 0; i < V_ITERS; ++i) vreg[i] = *(const u32x4*)(Vh + v_goff[i] + kt);
    }
  };
  auto store_tile = [&](int buf) {
    if constexpr (!DMA) {
      char* base = lds + buf * BUF;
#pragma unroll
      for (int i = 0; i < K_ITERS; ++i) *(u32x4*)(base + k_loff[i]) = kreg[i];
#pragma unroll
      for (int i = 0; i < V_ITERS; ++i) *(u32x4*)(base + v_loff[i]) = vreg[i];
    }
  };

  // oacc[qb][ND] is the softmax denominator: P is multiplied by one extra "V column" of ones (a register
  // constant: B-operand lanes of column 0 hold 1.0), so the row sums come out of the MFMA pipe in the same
  // layout and with the same rescaling as O instead of costing one VALU add per score.  The denominator
  // therefore sums the bf16-rounded probabilities - exactly the weights P*V uses.
  // head_dim 80 (VGPR-bound at three workgroups per CU) keeps the denominators as two f32 lane sums instead: that frees
  // the ones fragment and the extra accumulator block (12 VGPRs), which is what lets the LDS-DMA source offsets stay in
  // registers for the whole kernel (no per-tile address arithmetic, no scratch).
  constexpr bool MFMA_SUM = (HD == 128);
  constexpr int NACC = ND + (MFMA_SUM ? 1 : 0);
  f32x4 oacc[2][NACC];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int nb = 0; nb < NACC; ++nb) oacc[qb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // -m of each q-block's query (the lane's), the C operand of the first QK^T k-step: see attn_vit32_kernel
  f32x4 negm[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
  float lsum[2] = {0.f, 0.f};      // !MFMA_SUM: sum over this lane's keys (4 h-rows are added at the end)
  const uint32_t one2 = (l15 == 0) ? 0x3f803f80u : 0u;
  const bf16x8 ones_frag = __builtin_bit_cast(bf16x8, (u32x4){one2, one2, one2, one2});

  if (nt > 0) {
    load_tile(kt_begin, 0);
    store_tile(0);
  }
  __syncthreads();

  int cur = 0;
  for (int t = 0; t < nt; ++t) {
    const int kt = kt_begin + t * 64;
    const bool more = (t + 1 < nt);
    if (more && !((ATT_PROBE & 1) && t >= 1)) load_tile(kt + 64, cur ^ 1);

    const bool active = !CAUSAL || (kt <= wq0 + wrows - 1);
    auto tile_body = [&](auto nq_tag) {
      constexpr int NQ = decltype(nq_tag)::value;
      const char* kb = lds + cur * BUF;
      const char* vb = kb + K_BYTES;
      // ---- S^T = K * (scale Q)^T - m: the first k-step takes -m (negm) as its C operand
      f32x4 sacc[4][NQ];
#pragma unroll
      for (int ds = 0; ds < DKF; ++ds) {
        bf16x8 kf[4];
#pragma unroll
        for (int kbk = 0; kbk < 4; ++kbk) {
          const int row = kbk * 16 + l15;
          const int c = ds * 4 + h;
          const int pc = (HD == 128) ? (c ^ l15) : c;   // d = 80: dense 160-byte rows need no swizzle for b128 (see below)
          kf[kbk] = *(const bf16x8*)(kb + row * KROW + pc * 16);
        }
#pragma unroll
        for (int kbk = 0; kbk < 4; ++kbk)
#pragma unroll
          for (int qb = 0; qb < NQ; ++qb) {
            if constexpr ((ATT_PROBE & 8) != 0) { sacc[kbk][qb] = (f32x4){kf[kbk][0], kf[kbk][1], qf[qb][ds][0], 1.0f}; continue; }
            sacc[kbk][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kbk], qf[qb][ds], ds == 0 ? negm[qb] : sacc[kbk][qb], 0, 0, 0);
          }
      }

      const bool need_mask = (kt < k0) || (kt + 64 > k1) || (CAUSAL && (kt + 63 > wq0));
      if constexpr (TAIL16) {   // dims 64..79: chunk 8 + (h >> 1), 8-byte half (h & 1)
        bf16x4s kt4[4];
#pragma unroll
        for (int kbk = 0; kbk < 4; ++kbk) {
          const int row = kbk * 16 + l15;
          const int pc = (DKF * 4 + (h >> 1)) ^ ((l15 >> 3) & 1);
          kt4[kbk] = __builtin_bit_cast(bf16x4s, *(const u32x2*)(kb + row * KROW + pc * 16 + (h & 1) * 8));
        }
#pragma unroll
        for (int kbk = 0; kbk < 4; ++kbk)
#pragma unroll
          for (int qb = 0; qb < NQ; ++qb)
            sacc[kbk][qb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(kt4[kbk], qt[qb], sacc[kbk][qb], 0, 0, 0);
      }
      // ---- online softmax (query on the lane, keys in registers), log2 domain: the accumulators hold s - m, so
      //      p = one raw v_exp_f32 per score; m is set by the first tile and then only moves when a tile's maximum
      //      exceeds it by more than ATT_LAZY (delta below, per lane = per query; the branch is per wave)
      float delta[NQ];
#pragma unroll
      for (int qb = 0; qb < NQ; ++qb) {
        const int q = wq0 + qb * 16 + l15;
        if (need_mask) {
          // element (kbk, r) is key kt + 4h + (16 kbk + r): valid iff lo <= 16 kbk + r < hi, with the
          // compile-time constant on one side of each compare (no per-element index arithmetic)
          const int kbase = kt + 4 * h;
          int lo = k0 - kbase;
          int hi = (CAUSAL ? min(k1, q + 1) : k1) - kbase;
          // opaque to the optimiser, so the compares/selects stay inside this (uniform, rarely taken) branch
          // instead of being speculated into every tile
          asm volatile("" : "+v"(lo), "+v"(hi));
#pragma unroll
          for (int kbk = 0; kbk < 4; ++kbk)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const bool ok = (lo <= 16 * kbk + r) && (hi > 16 * kbk + r);
              sacc[kbk][qb][r] = ok ? sacc[kbk][qb][r] : ATT_NEG;
            }
        }
        // 8 x v_max3_f32 through att_max3 (inline asm): plain fmaxf on MFMA results makes hipcc put a canonicalising
        // v_max_f32 x, x, x in front of every chain
        float mx = att_max3(sacc[0][qb][0], sacc[0][qb][1], sacc[0][qb][2]);
        mx = att_max3(mx, sacc[0][qb][3], sacc[1][qb][0]);
        mx = att_max3(mx, sacc[1][qb][1], sacc[1][qb][2]);
        mx = att_max3(mx, sacc[1][qb][3], sacc[2][qb][0]);
        mx = att_max3(mx, sacc[2][qb][1], sacc[2][qb][2]);
        mx = att_max3(mx, sacc[2][qb][3], sacc[3][qb][0]);
        mx = att_max3(mx, sacc[3][qb][1], sacc[3][qb][2]);
        mx = att_max3(mx, sacc[3][qb][3], sacc[3][qb][3]);
        mx = att_hmax4(mx);  // over the four 16-lane rows (keys 4h..4h+3): two permlane swaps, no LDS round trip
        delta[qb] = (t == 0 || mx > ATT_LAZY) ? mx : 0.f;
      }
      // ---- a reference moved: shift this tile's scores and -m, rescale O (rows 4h+r of each q-block) and the lane sums
      if (!__all((delta[0] == 0.f) && (delta[NQ - 1] == 0.f))) {
        asm volatile("" : "+v"(delta[0]));  // keep this a real (rarely taken) branch: nothing of it is speculated
#pragma unroll
        for (int qb = 0; qb < NQ; ++qb) {
          const float d = delta[qb];
          const float nm = negm[qb][0] - d;
          negm[qb] = (f32x4){nm, nm, nm, nm};
#pragma unroll
          for (int kbk = 0; kbk < 4; ++kbk)
#pragma unroll
            for (int r = 0; r < 4; ++r) sacc[kbk][qb][r] -= d;
          const float alpha = att_exp2(-d);
          if constexpr (!MFMA_SUM) lsum[qb] *= alpha;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float a = __shfl(alpha, 4 * h + r, 64);
#pragma unroll
            for (int nb = 0; nb < NACC; ++nb) oacc[qb][nb][r] *= a;
          }
        }
      }
      bf16x8 pf[2][NQ];
#pragma unroll
      for (int qb = 0; qb < NQ; ++qb) {
        float pv[4][4];
#pragma unroll
        for (int kbk = 0; kbk < 4; ++kbk)
#pragma unroll
          for (int r = 0; r < 4; ++r) pv[kbk][r] = att_exp2(sacc[kbk][qb][r]);
        if constexpr (!MFMA_SUM) {
          float t0 = (pv[0][0] + pv[0][1]) + (pv[0][2] + pv[0][3]), t1 = (pv[1][0] + pv[1][1]) + (pv[1][2] + pv[1][3]);
          float t2 = (pv[2][0] + pv[2][1]) + (pv[2][2] + pv[2][3]), t3 = (pv[3][0] + pv[3][1]) + (pv[3][2] + pv[3][3]);
          lsum[qb] += (t0 + t1) + (t2 + t3);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          u32x4 pk;
          pk[0] = pack2bf(pv[2 * ks][0], pv[2 * ks][1]);
          pk[1] = pack2bf(pv[2 * ks][2], pv[2 * ks][3]);
          pk[2] = pack2bf(pv[2 * ks + 1][0], pv[2 * ks + 1][1]);
          pk[3] = pack2bf(pv[2 * ks + 1][2], pv[2 * ks + 1][3]);
          pf[ks][qb] = __builtin_bit_cast(bf16x8, pk);
        }
      }

      // ---- O += P * V  (B operand: chunk 4 ks + h of V^T row d, keys already in k-slot order)
      const int vsw = (l15 >> 1) & 7;
#pragma unroll
      for (int nb = 0; nb < ND; ++nb) {
        const char* vrow = vb + (nb * 16 + l15) * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const bf16x8 vf = *(const bf16x8*)(vrow + (((4 * ks + h) ^ vsw) << 4));
#pragma unroll
          for (int qb = 0; qb < NQ; ++qb) {
            if constexpr ((ATT_PROBE & 4) != 0) { oacc[qb][nb][0] += (float)vf[0] + (float)pf[ks][qb][0]; continue; }
            oacc[qb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[ks][qb], vf, oacc[qb][nb], 0, 0, 0);
          }
        }
      }
      if constexpr (MFMA_SUM) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int qb = 0; qb < NQ; ++qb)
            oacc[qb][ND] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[ks][qb], ones_frag, oacc[qb][ND], 0, 0, 0);
      }
    };
    if (active) {
      if (HD != 80 || nqb == 2) tile_body(std::integral_constant<int, 2>{});
      else tile_body(std::integral_constant<int, 1>{});
    }

    if (more) store_tile(cur ^ 1);
    if constexpr ((ATT_PROBE & 16) == 0) __syncthreads();
    cur ^= 1;
  }

  // ---- normalise, stage the wave's 32 x HD tile in LDS, store whole chunks
  char* ost = lds + wave * 32 * OROW;
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    if (qb >= nqb) break;
    float ltot = 0.f;
    if constexpr (!MFMA_SUM) ltot = hsum4(lsum[qb]);          // query l15's denominator, on all four of its lanes
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      // MFMA_SUM: column 0 of the ones block holds row 4h+r's sum; otherwise fetch query (4h + r)'s total from lane 4h+r
      const float l = MFMA_SUM ? __shfl(oacc[qb][MFMA_SUM ? ND : 0][r], 16 * h, 64) : __shfl(ltot, 4 * h + r, 64);
      const float a = (l > 0.f) ? 1.0f / l : 0.f;
      const int row = qb * 16 + 4 * h + r;
#pragma unroll
      for (int nb = 0; nb < ND; ++nb)
        *(bf16_t*)(ost + row * OROW + (nb * 16 + l15) * 2) = f2bf(oacc[qb][nb][r] * a);
    }
  }
  __syncthreads();
  for (int it = lane; it < wrows * KCH; it += 64) {
    const int row = it / KCH, c = it - row * KCH;
    const int q = wq0 + row;
    if (q < q0 + qn) {
      const u32x4 o = *(const u32x4*)(ost + row * OROW + c * 16);
      *(u32x4*)(r_O + (size_t)(q - p.q_row0) * p.ldo + head * HD + c * 8) = o;
    }
  }
}

// ---------------------------------------------------------------------------
// K6 (r03): the ViT kernel - head_dim 80, non-causal - on v_mfma_f32_32x32x16_bf16.
//
// Why another MFMA shape: flash attention at head_dim 80 issues ~11 vector instructions per MFMA of the 16-wide form
// (fma + exp + max + add + pack per score, 16 leftover dims on a third MFMA form); the 32-wide form needs 22 MFMAs and
// ~100 vector instructions per (32 query x 64 key) wave tile instead of 40 and ~330, has no tail form, and its softmax
// denominator is free (below).  Here a wave owns ONE 32-row query block:
//   S^T[key][q] = K * Q^T      A = K rows from LDS (32 keys x 16 d per MFMA, one ds_read_b128), B = Q^T in registers;
//                              head_dim 80 = 5 k-steps of 16, no tail form; 10 MFMAs per 64-key tile
//   softmax                    the query is the lane (lane & 31), its 64 keys of the tile are 32 registers here and 32 on
//                              lane ^ 32: row max = 10 v_max3 + ONE permlane32 swap; alpha / 1/l are lane-local
//   O^T[d][q] += V^T * P^T     A = V^T rows from LDS (32 d x 16 keys, one ds_read_b128), B = the exponentiated S^T
//                              registers themselves (v_cvt_pk only; the k-slot order is matched by WHICH accumulator
//                              registers form a fragment: registers 4t..4t+3 and 4t+8..4t+11 of a 32-key block = the keys of
//                              V^T chunk 2t + lane half, in the column order vis_qkv_rope_split already writes);
//                              d = 80 pads to three 32-row blocks: 12 MFMAs per tile instead of 10 - and the pad rows pay
//                              for themselves: LDS row 80 of the V^T image holds ones, so O^T[80][q] IS the softmax
//                              denominator (sum of the bf16-rounded probabilities, exactly the weights P*V uses) - no
//                              v_add per score, no extra MFMA.
// Measured (r03, 4900 patches x 16 heads, N(0,1) data): 192-195 us against 203-214 us for the 16-wide kernel.  PMC
// (profiles/r03_vit_attention_pmc.txt): MFMA pipe 44 % busy, vector instructions active 50 % of the SIMD's time, and the
// two never overlap beyond the MFMA's own issue cycles (SQ_VALU_MFMA_COEXEC_CYCLES = 8 x the MFMA count): on this part a
// SIMD's time is MFMA-busy + vector-issue cycles, whichever wave they come from.  Two restructurings that try to hide one
// behind the other - a 12-wave workgroup whose three wave groups run QK^T / softmax / P*V as a three-stage pipeline, and a
// hand-placed MFMA / VALU stream inside each wave at 32-key half tiles - are correct and land on the same time
// (tools/probes/attn_vit_variants.hip, vit_probe.py: stamps show a ~100-instruction softmax phase taking 700 cycles alone
// and 1700 beside two MFMA waves, with or without s_setprio).  They are kept as probe sources, not in this library.
// LDS images: K dense 160-byte rows, 16-byte chunk index XOR ((row >> 3) & 1) - under the ds_read_b128 lane groups
// ({0-3,12-15,20-27}, ...) with 32 rows per instruction this is the conflict-free form (10 r mod 16 repeats with period
// 8; the XOR moves rows 8..15 / 24..31 to the odd slots); V^T 128-byte rows, chunk XOR ((d >> 1) & 7), as above.  Both
// arrive by LDS-DMA with the swizzle on the source address; slots past row 63 / 79 are not written at all (waves 2, 3
// skip their third piece), the pad rows 80..95 are set once.
// Work items, masks, output staging and the fixed key-tile grid (absolute 64-key tiles: results do not depend on what
// shares the launch) are those of attn_prefill_kernel; items of <= 64 rows keep two of the four waves busy.
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define ATT_SPLIT_ROW 84          // 128-float rows of a key-split partial ([d][query row]): O^T[0..79], l, m, 2 unused

__global__ __launch_bounds__(256, 3) void attn_vit32_kernel(AttnArgs p) {
  constexpr int HD = 80, KS = HD / 16, NDB = 3;
  constexpr int K_BYTES = 12288, V_BYTES = 12288, BUF = K_BYTES + V_BYTES;   // 768 + 768 16-byte slots
  constexpr int OROW = HD * 2 + 16;
  static_assert(2 * BUF >= 4 * 32 * OROW, "O staging must fit in the KV buffers");
  __shared__ __attribute__((aligned(16))) char lds[2 * BUF];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r31 = lane & 31, hh = lane >> 5;
  const int head = blockIdx.x, hkv = head / p.group;
  const int4 wk = p.work[blockIdx.y];
  const int q0 = wk.x, qn = wk.y & 0xff, k0 = wk.z, k1 = wk.w;
  const int split = wk.y >> 8;          // 0, or 1 | part << 1 | pair << 2 (ATT_SPLIT_* below)
  const int kt_begin = k0 & ~63;
  const int nt = (k1 - kt_begin + 63) >> 6;
  const int wq0 = q0 + wave * 32;
  const bool active = wave * 32 < qn;

  const bf16_t* Kh = p.K + (size_t)hkv * p.k_tokens * HD;
  const bf16_t* Vh = p.Vt + (size_t)hkv * HD * p.vt_ld;

  // pad rows 80..95 of both V^T images: row 80 = 1.0 in every column, rows 81..95 = 0 (never overwritten by the DMA)
  {
    const int b = tid >> 7, slot = tid & 127;
    const uint32_t v = (slot < 8) ? 0x3f803f80u : 0u;
    *(u32x4*)(lds + b * BUF + K_BYTES + 80 * 128 + slot * 16) = (u32x4){v, v, v, v};
  }

  // Q^T fragments (B operand): lane (q = r31, hh) holds Q[q][16 ks + 8 hh .. + 7]
  bf16x8 qf[KS];
  {
    const int qrow = min(wq0 + r31, p.q_row0 + p.Sq - 1) - p.q_row0;
    const bf16_t* qp = p.Q + ((size_t)head * p.Sq + qrow) * HD + 8 * hh;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      qf[ks] = att_scaled_q8(qp + ks * 16, p.scale_log2);
    }
  }

  // LDS-DMA source offsets (loop constants): slot ps = i * 256 + tid; the third piece exists for ps < 640 only
  uint32_t dk_off[3], dv_off[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int ps = min(i * 256 + tid, 639);
    const int row = (ps * 6554) >> 16;                  // ps / 10 for ps < 768
    const int c = ps - row * 10;
    dk_off[i] = (uint32_t)row * (HD * 2) + ((c ^ ((row >> 3) & 1)) << 4);
    const int d = ps >> 3, cv = ps & 7;
    dv_off[i] = (uint32_t)d * (uint32_t)(p.vt_ld * 2) + ((cv ^ ((d >> 1) & 7)) << 4);
  }
  const int dma_base = wave * 1024;
  // LDS-DMA through BUFFER instructions (r05): the per-lane offsets above are loop constants in VGPRs, the tile's position is
  // the instruction's SCALAR offset, the head's base lives in the resource descriptor - no per-tile vector address arithmetic
  // (the global_load_lds form rebuilt a 64-bit address per piece and tile: 14 v_lshl_add_u64 + ~40 other VALU / SALU
  // instructions per tile, profiles/r05_vit_attention_isa.txt), and rows past the head's last key read as zeros by the
  // descriptor's range check (the ragged last tile needed a second code path with clamped rows; its scores are masked anyway).
  const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc((void*)Kh, 0, p.k_tokens * (HD * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc((void*)Vh, 0, HD * p.vt_ld * 2, 0x00020000);
  auto load_tile = [&](int kt, int buf) {
    __attribute__((address_space(3))) char* base = (__attribute__((address_space(3))) char*)(lds + buf * BUF + dma_base);
    const int ko = kt * (HD * 2), vo = kt * 2;
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (i < 2 || wave < 2)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_k, (__attribute__((address_space(3))) void*)(base + i * 4096), 16, dk_off[i], ko, 0, 0);
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (i < 2 || wave < 2)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_v, (__attribute__((address_space(3))) void*)(base + K_BYTES + i * 4096), 16, dv_off[i], vo, 0, 0);
  };

  // fragment read addresses: one per-lane base for K (the chunk XOR only touches bit 0: (2 ks + hh) ^ sw = 2 ks + (hh ^ sw)),
  // four for V^T (chunk 4 kb + 2 t + hh XOR (d >> 1) & 7); everything else is an immediate
  const int k_lane = r31 * (HD * 2) + ((hh ^ ((r31 >> 3) & 1)) << 4);
  int v_lane[4];
#pragma unroll
  for (int c2 = 0; c2 < 4; ++c2) v_lane[c2] = K_BYTES + r31 * 128 + ((((2 * c2) | hh) ^ ((r31 >> 1) & 7)) << 4);

  f32x16 oacc[NDB];
#pragma unroll
  for (int db = 0; db < NDB; ++db)
#pragma unroll
    for (int i = 0; i < 16; ++i) oacc[db][i] = 0.f;
  // Reference value m of the online softmax, kept as the MFMA's C operand: S^T = K * (scale Q)^T + (-m) leaves
  // s - m in the accumulators, so a score costs v_exp + half a v_cvt_pk and nothing else.  m is set by the first tile
  // and afterwards only moves when a tile's maximum exceeds it by more than ATT_LAZY (above): the common
  // tile has no alpha, no rescale and no per-score subtraction; the rare one pays them for the whole wave.
  f32x16 negm;
#pragma unroll
  for (int i = 0; i < 16; ++i) negm[i] = 0.f;

  if (nt > 0) load_tile(kt_begin, 0);
  __syncthreads();

  int cur = 0;
  for (int t = 0; t < nt; ++t) {
    const int kt = kt_begin + t * 64;
    const bool more = (t + 1 < nt);
    if (more) load_tile(kt + 64, cur ^ 1);

    if (active) {
      const char* kb_ = lds + cur * BUF;
      // ---- S^T = K * Q^T : two 32-key blocks
      f32x16 sacc[2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        bf16x8 kf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kf[ks] = *(const bf16x8*)(kb_ + k_lane + kb * (32 * HD * 2) + ks * 32);
        f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qf[0], negm, 0, 0, 0);
#pragma unroll
        for (int ks = 1; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], acc, 0, 0, 0);
        sacc[kb] = acc;
      }
      // ---- mask the edge tiles: element (kb, i) is key kt + 4 hh + 32 kb + 8 (i >> 2) + (i & 3)
      const bool need_mask = (kt < k0) || (kt + 64 > k1);
      if (need_mask) {
        const int kbase = kt + 4 * hh;
        int lo = k0 - kbase, hi = k1 - kbase;
        asm volatile("" : "+v"(lo), "+v"(hi));      // keep the compares inside this (uniform, rarely taken) branch
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int e = 32 * kb + 8 * (i >> 2) + (i & 3);
            sacc[kb][i] = ((lo <= e) && (hi > e)) ? sacc[kb][i] : ATT_NEG;
          }
      }
      // ---- online softmax, log2 domain; the other 32 keys of this query are on lane ^ 32
      float mx = att_max3(sacc[0][0], sacc[0][1], sacc[0][2]);
#pragma unroll
      for (int i = 3; i < 15; i += 2) mx = att_max3(mx, sacc[0][i], sacc[0][i + 1]);
      mx = att_max3(mx, sacc[0][15], sacc[1][0]);
#pragma unroll
      for (int i = 1; i < 15; i += 2) mx = att_max3(mx, sacc[1][i], sacc[1][i + 1]);
      mx = att_max(mx, sacc[1][15]);
      {
        const uint32_t u = __float_as_uint(mx);
        const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        mx = att_max(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
      }
      // mx = max(s - m) of this query over the tile.  delta = how far m moves: the whole maximum on the first tile,
      // the excess when it is more than ATT_LAZY above m, else 0 (per lane = per query; the branch is per wave)
      const float delta = (t == 0 || mx > ATT_LAZY) ? mx : 0.f;
      if (!__all(delta == 0.f)) {
        float d = delta;
        asm volatile("" : "+v"(d));                 // a real (rarely taken) branch: nothing of it is speculated
        const float nm = negm[0] - d;
#pragma unroll
        for (int i = 0; i < 16; ++i) negm[i] = nm;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int i = 0; i < 16; ++i) sacc[kb][i] -= d;
        const float a = att_exp2(-d);               // rescale O^T (the query is the lane: no cross-lane traffic)
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
          for (int i = 0; i < 16; ++i) oacc[db][i] *= a;
      }
      bf16x8 pf[2][2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        float e[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) e[i] = att_exp2(sacc[kb][i]);
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
          u32x4 pk;
          pk[0] = pack2bf(e[4 * t2], e[4 * t2 + 1]);
          pk[1] = pack2bf(e[4 * t2 + 2], e[4 * t2 + 3]);
          pk[2] = pack2bf(e[4 * t2 + 8], e[4 * t2 + 9]);
          pk[3] = pack2bf(e[4 * t2 + 10], e[4 * t2 + 11]);
          pf[kb][t2] = __builtin_bit_cast(bf16x8, pk);
        }
      }
      // ---- O^T += V^T * P^T
#pragma unroll
      for (int db = 0; db < NDB; ++db) {
        bf16x8 vf[4];
#pragma unroll
        for (int c2 = 0; c2 < 4; ++c2) vf[c2] = *(const bf16x8*)(kb_ + v_lane[c2] + db * (32 * 128));
#pragma unroll
        for (int c2 = 0; c2 < 4; ++c2)
          oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[c2], pf[c2 >> 1][c2 & 1], oacc[db], 0, 0, 0);
      }
    }
    __syncthreads();
    cur ^= 1;
  }

  // ---- key-split item: this workgroup saw only half of the keys.  It leaves its unnormalised O^T, l and m in the
  //      pair's slot; the LATER of the two workgroups (an agent-scope counter tells which) adds the other half to its
  //      own registers - a_self * x_self + a_other * x_other with separately rounded products, so the sum does not
  //      depend on which half arrived last - and goes on to the ordinary epilogue.  Nobody waits for anybody.
  if (split && p.ws_part) {   // (flagged items through an entry point without a workspace are treated as whole items)
    const int part = (split >> 1) & 1, pair = min(split >> 2, p.n_pairs - 1);
    const size_t slot = (size_t)pair * gridDim.x + head;
    float* mine = p.ws_part + (slot * 2 + part) * (size_t)(128 * ATT_SPLIT_ROW);
    const float* other = p.ws_part + (slot * 2 + (part ^ 1)) * (size_t)(128 * ATT_SPLIT_ROW);
    const int row = wave * 32 + r31;
    const float m_self = -negm[0];
    // Visibility without flushing caches the kernel lives on: the partials travel as relaxed AGENT-scope atomic stores /
    // loads (write-through / coherent reads of exactly these words), the writer orders them before its counter update
    // with a release fence (L2 write-back of the few dirty lines there are), and the reader takes NO acquire fence - an
    // acquire invalidates the whole L2 of the XCD, K / V tiles of every other workgroup included (measured: 576 of
    // them per launch made the split launch slower than the unsplit one).
    if (active) {
#pragma unroll
      for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if (db == 2 && g >= 2) break;
          const int d0 = 32 * db + 8 * g + 4 * hh;
#pragma unroll
          for (int j = 0; j < 4; ++j)     // [d][row]: 32 lanes = 128 contiguous bytes per store
            __hip_atomic_store(mine + (d0 + j) * 128 + row, oacc[db][4 * g + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      if (hh == 0) {
        __hip_atomic_store(mine + 80 * 128 + row, oacc[2][8], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(mine + 81 * 128 + row, m_self, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
#ifdef ATT_SPLIT_FENCE
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the sc1 stores above are acknowledged = written through
#endif
    __syncthreads();                       // every thread's partial is out before the counter moves
    int* flag = (int*)lds;
    if (tid == 0) *flag = __hip_atomic_fetch_add(p.ws_count + slot, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int arrived = *flag;
    if (arrived == 0) return;              // the partner has not finished: it will merge
    asm volatile("" ::: "memory");
    if (tid == 0) __hip_atomic_store(p.ws_count + slot, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    if (active) {
#pragma clang fp contract(off)             // a_s * x_s + a_o * x_o must not become an FMA: the sum has to be symmetric
      const float l_other = __hip_atomic_load(other + 80 * 128 + row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const float m_other = __hip_atomic_load(other + 81 * 128 + row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const float m = fmaxf(m_self, m_other);
      const float a_s = att_exp2(m_self - m), a_o = att_exp2(m_other - m);
#pragma unroll
      for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if (db == 2 && g >= 2) break;
          const int d0 = 32 * db + 8 * g + 4 * hh;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float x = __hip_atomic_load(other + (d0 + j) * 128 + row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float ts = a_s * oacc[db][4 * g + j];
            const float to = a_o * x;
            oacc[db][4 * g + j] = ts + to;
          }
        }
      const float ls = a_s * oacc[2][8], lo = a_o * l_other;
      oacc[2][8] = ls + lo;                // l (meaningful on the lower half)
    }
    __syncthreads();                       // the flag word is part of the O staging area
  }

  // ---- normalise (denominator = O^T row 80 = register 8 of block 2 on the lower lane half), stage the wave's
  //      32 x 80 tile in LDS, store whole 16-byte chunks
  char* ost = lds + wave * 32 * OROW;
  if (active) {
    const uint32_t lu = __float_as_uint(oacc[2][8]);
    const auto sw = __builtin_amdgcn_permlane32_swap(lu, lu, false, false);
    const float l = __uint_as_float(sw[0]);          // the lower half's value on both halves
    const float a = (l > 0.f) ? 1.0f / l : 0.f;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (db == 2 && g >= 2) break;                // d >= 80: pad rows
        const int d0 = 32 * db + 8 * g + 4 * hh;
        u32x2 o2;
        o2[0] = pack2bf(oacc[db][4 * g] * a, oacc[db][4 * g + 1] * a);
        o2[1] = pack2bf(oacc[db][4 * g + 2] * a, oacc[db][4 * g + 3] * a);
        *(u32x2*)(ost + r31 * OROW + d0 * 2) = o2;
      }
  }
  __syncthreads();
  if (active) {
    for (int it = lane; it < 32 * 10; it += 64) {
      const int row = (it * 6554) >> 16, c = it - row * 10;
      const int q = wq0 + row;
      if (q < q0 + qn) {
        const u32x4 o = *(const u32x4*)(ost + row * OROW + c * 16);
        *(u32x4*)(p.O + (size_t)(q - p.q_row0) * p.ldo + head * HD + c * 8) = o;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Causal prefill, head_dim 128, BALANCED: one workgroup = 8 waves = a PAIR of 128-row query blocks, a late ("heavy")
// block B and an early ("light") block A of the same sequence.  With one block per workgroup (the kernel above) the
// causal grid is a single round of the chip whose duration is the LAST block's (S = 2249: 36 key tiles, while the
// average block has 18) - half of the chip's time is idle.  Here wave w owns 16 rows of B (its q-block 0) AND 16 rows
// of A (its q-block 1): while the key tile is still below A's diagonal every wave runs the two-block tile body, after
// it the one-block body on B alone - every wave of every workgroup does the same number of (16-row block x key tile)
// units (pairs (last, first), (last-1, second) ...), and the K / V^T tile is staged once for both blocks.
// Work item {qB0, qBn, qA0, qAn} (qAn == 0: no partner); keys [0, min(k_tokens, q + 1)); rows are sequence positions,
// Q / O hold rows q_row0 .. q_row0 + Sq - 1 as in vis_attn_prefill_rows.  Same MFMA formulation, LDS images, online
// softmax and bf16 rounding points as attn_prefill_kernel<128, true>; the summation order over key tiles is the same,
// so a row's result is bit-identical to the unpaired kernel's.
__global__ __launch_bounds__(512, 2) void attn_prefill_pair_kernel(AttnArgs p) {
  constexpr int HD = 128, DKF = 4, ND = 8, KCH = 16, KROW = 256;
  constexpr int K_BYTES = 64 * KROW, V_BYTES = HD * 128, BUF = K_BYTES + V_BYTES;   // 16 KiB + 16 KiB
  constexpr int OROW = HD * 2 + 16;
  static_assert(2 * BUF >= 8 * 16 * OROW, "O staging (16 rows per wave per pass) must fit in the KV buffers");
  __shared__ __attribute__((aligned(16))) char lds[2 * BUF];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, h = lane >> 4;
  const int head = blockIdx.x, hkv = head / p.group;
  const int4 wk = p.work[blockIdx.y];
  // (locals, not updates of `p`: writing to a by-value kernel argument makes the compiler keep the whole struct in scratch)
  const int z = blockIdx.z;
  const bf16_t* const r_Q = p.Q + z * p.req.q_bs;
  const bf16_t* const r_K = p.K + req_kv(p.req, z);
  const bf16_t* const r_Vt = p.Vt + z * p.req.vt_bs;
  bf16_t* const r_O = p.O + z * p.req.o_bs;
  // q-block 0 = heavy block B, q-block 1 = light block A
  const int qblk0[2] = {wk.x, wk.z}, qblkn[2] = {wk.y, wk.w};
  const int kend[2] = {min(p.k_tokens, wk.x + wk.y), min(p.k_tokens, wk.z + wk.w)};   // one past the last key of a block
  const int nt = (kend[0] + 63) >> 6;
  const int wrow0[2] = {wk.x + wave * 16, wk.z + wave * 16};            // first row of this wave in each block
  const bool have[2] = {wave * 16 < wk.y, wk.w > 0 && wave * 16 < wk.w};

  const bf16_t* Kh = r_K + (size_t)hkv * p.k_tokens * HD;
  const bf16_t* Vh = r_Vt + (size_t)hkv * HD * p.vt_ld;

  bf16x8 qf[2][DKF];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int qrow = min(max(wrow0[qb] + l15, p.q_row0), p.q_row0 + p.Sq - 1) - p.q_row0;
    const bf16_t* qp = r_Q + ((size_t)head * p.Sq + qrow) * HD;
#pragma unroll
    for (int ds = 0; ds < DKF; ++ds) qf[qb][ds] = att_scaled_q8(qp + ds * 32 + 8 * h, p.scale_log2);
  }

  // staging (r05: LDS-DMA through buffer instructions, as attn_vit32_kernel; r02-r04 went global -> registers -> ds_write, the
  // loads of tile t + 1 parked in 16 VGPRs over the whole body of tile t and written behind it): K tile 64 rows x 16 chunks,
  // V^T tile 128 rows x 8 chunks = 1024 16-byte slots each = two 8 KiB pieces per image (512 threads x 16 B).  The LDS images
  // are lane-linear, so the XOR swizzles sit on the per-lane SOURCE offsets (loop constants); the tile's position is the
  // instruction's scalar offset; rows past the head's last key read as zeros by the descriptor's range check (masked below).
  uint32_t dk_off[2], dv_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int it = tid + i * 512;
    const int row = it >> 4, c = it & 15;
    dk_off[i] = (uint32_t)row * (HD * 2) + ((c ^ (row & 15)) << 4);
    const int d = it >> 3, cv = it & 7;
    dv_off[i] = (uint32_t)d * (uint32_t)(p.vt_ld * 2) + ((cv ^ ((d >> 1) & 7)) << 4);
  }
  const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc((void*)Kh, 0, p.k_tokens * (HD * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc((void*)Vh, 0, HD * p.vt_ld * 2, 0x00020000);
  const int dma_base = __builtin_amdgcn_readfirstlane(wave) * 1024;
  auto dma_tile = [&](int kt, int buf) {
    __attribute__((address_space(3))) char* base = (__attribute__((address_space(3))) char*)(lds + buf * BUF + dma_base);
    const int ko = kt * (HD * 2), vo = kt * 2;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_k, (__attribute__((address_space(3))) void*)(base + i * 8192), 16, dk_off[i], ko, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_v, (__attribute__((address_space(3))) void*)(base + K_BYTES + i * 8192), 16, dv_off[i], vo, 0, 0);
  };

  f32x4 oacc[2][ND + 1];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int nb = 0; nb <= ND; ++nb) oacc[qb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 negm[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};   // as in attn_prefill_kernel
  const uint32_t one2 = (l15 == 0) ? 0x3f803f80u : 0u;
  const bf16x8 ones_frag = __builtin_bit_cast(bf16x8, (u32x4){one2, one2, one2, one2});

  if (nt > 0) dma_tile(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int cur = 0;
  for (int t = 0; t < nt; ++t) {
    const int kt = t * 64;
    const bool more = (t + 1 < nt);
    if (more) dma_tile(kt + 64, cur ^ 1);   // the other buffer: last read in iteration t - 1, behind that iteration's barrier

    // a 16-row block of this wave is live while the tile starts at or below its last row
    const bool live0 = have[0] && kt <= wrow0[0] + 15;
    const bool live1 = have[1] && kt <= wrow0[1] + 15;
    auto tile_body = [&](auto nq_tag) {
      constexpr int NQ = decltype(nq_tag)::value;
      const char* kb = lds + cur * BUF;
      const char* vb = kb + K_BYTES;
      f32x4 sacc[4][NQ];
#pragma unroll
      for (int ds = 0; ds < DKF; ++ds) {
        bf16x8 kf[4];
#pragma unroll
        for (int kbk = 0; kbk < 4; ++kbk) {
          const int row = kbk * 16 + l15;
          kf[kbk] = *(const bf16x8*)(kb + row * KROW + (((ds * 4 + h) ^ l15) << 4));
        }
#pragma unroll
        for (int kbk = 0; kbk < 4; ++kbk)
#pragma unroll
          for (int qb = 0; qb < NQ; ++qb)
            sacc[kbk][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kbk], qf[qb][ds], ds == 0 ? negm[qb] : sacc[kbk][qb], 0, 0, 0);
      }
      float delta[NQ];
#pragma unroll
      for (int qb = 0; qb < NQ; ++qb) {
        const int q = wrow0[qb] + l15;
        const bool need_mask = (kt + 63 > wrow0[qb]) || (kt + 64 > kend[qb]);
        if (need_mask) {
          const int kbase = kt + 4 * h;
          int hi = min(kend[qb], q + 1) - kbase;
          asm volatile("" : "+v"(hi));
#pragma unroll
          for (int kbk = 0; kbk < 4; ++kbk)
#pragma unroll
            for (int r = 0; r < 4; ++r) sacc[kbk][qb][r] = (hi > 16 * kbk + r) ? sacc[kbk][qb][r] : ATT_NEG;
        }
        float mx = att_max3(sacc[0][qb][0], sacc[0][qb][1], sacc[0][qb][2]);
        mx = att_max3(mx, sacc[0][qb][3], sacc[1][qb][0]);
        mx = att_max3(mx, sacc[1][qb][1], sacc[1][qb][2]);
        mx = att_max3(mx, sacc[1][qb][3], sacc[2][qb][0]);
        mx = att_max3(mx, sacc[2][qb][1], sacc[2][qb][2]);
        mx = att_max3(mx, sacc[2][qb][3], sacc[3][qb][0]);
        mx = att_max3(mx, sacc[3][qb][1], sacc[3][qb][2]);
        mx = att_max3(mx, sacc[3][qb][3], sacc[3][qb][3]);
        mx = att_hmax4(mx);
        delta[qb] = (t == 0 || mx > ATT_LAZY) ? mx : 0.f;
      }
      if (!__all((delta[0] == 0.f) && (delta[NQ - 1] == 0.f))) {
        asm volatile("" : "+v"(delta[0]));
#pragma unroll
        for (int qb = 0; qb < NQ; ++qb) {
          const float d = delta[qb];
          const float nm = negm[qb][0] - d;
          negm[qb] = (f32x4){nm, nm, nm, nm};
#pragma unroll
          for (int kbk = 0; kbk < 4; ++kbk)
#pragma unroll
            for (int r = 0; r < 4; ++r) sacc[kbk][qb][r] -= d;
          const float alpha = att_exp2(-d);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float a = __shfl(alpha, 4 * h + r, 64);
#pragma unroll
            for (int nb = 0; nb <= ND; ++nb) oacc[qb][nb][r] *= a;
          }
        }
      }
      bf16x8 pf[2][NQ];
#pragma unroll
      for (int qb = 0; qb < NQ; ++qb) {
        float pv[4][4];
#pragma unroll
        for (int kbk = 0; kbk < 4; ++kbk)
#pragma unroll
          for (int r = 0; r < 4; ++r) pv[kbk][r] = att_exp2(sacc[kbk][qb][r]);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          u32x4 pk;
          pk[0] = pack2bf(pv[2 * ks][0], pv[2 * ks][1]);
          pk[1] = pack2bf(pv[2 * ks][2], pv[2 * ks][3]);
          pk[2] = pack2bf(pv[2 * ks + 1][0], pv[2 * ks + 1][1]);
          pk[3] = pack2bf(pv[2 * ks + 1][2], pv[2 * ks + 1][3]);
          pf[ks][qb] = __builtin_bit_cast(bf16x8, pk);
        }
      }
      const int vsw = (l15 >> 1) & 7;
#pragma unroll
      for (int nb = 0; nb < ND; ++nb) {
        const char* vrow = vb + (nb * 16 + l15) * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const bf16x8 vf = *(const bf16x8*)(vrow + (((4 * ks + h) ^ vsw) << 4));
#pragma unroll
          for (int qb = 0; qb < NQ; ++qb)
            oacc[qb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[ks][qb], vf, oacc[qb][nb], 0, 0, 0);
        }
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int qb = 0; qb < NQ; ++qb)
          oacc[qb][ND] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[ks][qb], ones_frag, oacc[qb][ND], 0, 0, 0);
    };
    if (live1) tile_body(std::integral_constant<int, 2>{});        // A live implies B live (A lies before B)
    else if (live0) tile_body(std::integral_constant<int, 1>{});

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the next tile have landed
    __syncthreads();
    cur ^= 1;
  }

  // ---- normalise and store: 16 rows per wave per pass through a wave-private staging area
  char* ost = lds + wave * 16 * OROW;
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    if (qb == 1) __syncthreads();          // (also keeps the two passes' LDS traffic apart)
    if (!have[qb]) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float l = __shfl(oacc[qb][ND][r], 16 * h, 64);
      const float a = (l > 0.f) ? 1.0f / l : 0.f;
      const int row = 4 * h + r;
#pragma unroll
      for (int nb = 0; nb < ND; ++nb)
        *(bf16_t*)(ost + row * OROW + (nb * 16 + l15) * 2) = f2bf(oacc[qb][nb][r] * a);
    }
    for (int it = lane; it < 16 * KCH; it += 64) {
      const int row = it >> 4, c = it & 15;
      const int q = wrow0[qb] + row;
      if (q < qblk0[qb] + qblkn[qb]) {
        const u32x4 o = *(const u32x4*)(ost + row * OROW + c * 16);
        *(u32x4*)(r_O + (size_t)(q - p.q_row0) * p.ldo + head * HD + c * 8) = o;
      }
    }
  }
}

// Paired causal prefill (head_dim 128): work = n_work x int4 {qB0, qBn, qA0, qAn}, see attn_prefill_pair_kernel.
extern "C" int vis_attn_prefill_pairs(const void* Q, const void* K, const void* Vt, void* O, const void* work,
                                      int n_work, int Hq, int Hkv, int HD, int Sq, int k_tokens, int vt_ld, int ldo,
                                      float scale, int q_row0, hipStream_t stream) {
  if (!Q || !K || !Vt || !O || !work || n_work <= 0 || Hq <= 0 || Hkv <= 0 || Hq % Hkv != 0 || q_row0 < 0) return VIS_ERR_ARG;
  if (HD != 128) return VIS_ERR_ARG;
  if (Sq <= 0 || k_tokens <= 0 || vt_ld % 64 != 0 || ldo % 8 != 0 || ldo < Hq * HD || n_work > 65535) return VIS_ERR_ARG;
  if (((uintptr_t)Q | (uintptr_t)K | (uintptr_t)Vt | (uintptr_t)O | (uintptr_t)work) & 15) return VIS_ERR_ARG;
  AttnArgs p;
  p.Q = (const bf16_t*)Q; p.K = (const bf16_t*)K; p.Vt = (const bf16_t*)Vt; p.O = (bf16_t*)O;
  p.work = (const int4*)work;
  p.Sq = Sq; p.k_tokens = k_tokens; p.vt_ld = vt_ld; p.ldo = ldo; p.group = Hq / Hkv;
  p.scale_log2 = scale * 1.4426950408889634f;
  p.q_row0 = q_row0;
  p.ws_count = nullptr; p.ws_part = nullptr; p.n_pairs = 0;
  req_offsets_none(p.req);
  vis_clear_error();
  hipLaunchKernelGGL(attn_prefill_pair_kernel, dim3(Hq, n_work), dim3(512), 0, stream, p);
  return vis_check_launch();
}

// vis_attn_prefill_pairs for the `nreq` (<= 8) requests of a prompt-pass group in ONE launch (same work list: the requests share
// their prompt structure): request r reads Q + r * q_bs, K + kv_off[r] (element offset of its cache slot; host array), Vt + r * vt_bs
// and writes O + r * o_bs.  Rows are computed exactly as by vis_attn_prefill_pairs; what changes is the grid: a suffix pass of 1289
// rows is 28 x 6 = 168 workgroups - a third of the 512 the device holds - and four of them in a row took 4 x 60 us.
extern "C" int vis_attn_prefill_pairs_many(const void* Q, const void* K, const void* Vt, void* O, const void* work, int n_work,
                                           int Hq, int Hkv, int HD, int Sq, int k_tokens, int vt_ld, int ldo, float scale,
                                           int q_row0, int nreq, long long q_bs, long long vt_bs, long long o_bs,
                                           const long long* kv_off, hipStream_t stream) {
  if (!Q || !K || !Vt || !O || !work || n_work <= 0 || Hq <= 0 || Hkv <= 0 || Hq % Hkv != 0 || q_row0 < 0) return VIS_ERR_ARG;
  if (HD != 128 || nreq < 1 || nreq > VIS_MAX_REQ || !kv_off) return VIS_ERR_ARG;
  if (Sq <= 0 || k_tokens <= 0 || vt_ld % 64 != 0 || ldo % 8 != 0 || ldo < Hq * HD || n_work > 65535) return VIS_ERR_ARG;
  if (((uintptr_t)Q | (uintptr_t)K | (uintptr_t)Vt | (uintptr_t)O | (uintptr_t)work) & 15) return VIS_ERR_ARG;
  if (q_bs < 0 || vt_bs < 0 || o_bs < 0 || (q_bs | vt_bs | o_bs) % 8) return VIS_ERR_ARG;
  AttnArgs p;
  p.Q = (const bf16_t*)Q; p.K = (const bf16_t*)K; p.Vt = (const bf16_t*)Vt; p.O = (bf16_t*)O;
  p.work = (const int4*)work;
  p.Sq = Sq; p.k_tokens = k_tokens; p.vt_ld = vt_ld; p.ldo = ldo; p.group = Hq / Hkv;
  p.scale_log2 = scale * 1.4426950408889634f;
  p.q_row0 = q_row0;
  p.ws_count = nullptr; p.ws_part = nullptr; p.n_pairs = 0;
  req_offsets_none(p.req);
  for (int r = 0; r < nreq; ++r) {
    if (kv_off[r] < 0 || kv_off[r] % 8) return VIS_ERR_ARG;
    p.req.kv[r] = kv_off[r];
  }
  p.req.q_bs = q_bs; p.req.vt_bs = vt_bs; p.req.o_bs = o_bs;
  vis_clear_error();
  hipLaunchKernelGGL(attn_prefill_pair_kernel, dim3(Hq, n_work, nreq), dim3(512), 0, stream, p);
  return vis_check_launch();
}

// vis_attn_prefill with a row offset: the work items' query rows (and the causal rule key <= query) are positions of
// the whole sequence, while Q / O hold only rows q_row0 .. q_row0 + Sq - 1 (the prompt pass of a request whose first
// q_row0 tokens were computed elsewhere: a text prefix shared by the images of a batch).
extern "C" int vis_attn_prefill_rows(const void* Q, const void* K, const void* Vt, void* O, const void* work,
                                     int n_work, int Hq, int Hkv, int HD, int Sq, int k_tokens, int vt_ld, int ldo,
                                     int causal, float scale, int q_row0, hipStream_t stream);

extern "C" int vis_attn_prefill(const void* Q, const void* K, const void* Vt, void* O, const void* work,
                                int n_work, int Hq, int Hkv, int HD, int Sq, int k_tokens, int vt_ld, int ldo,
                                int causal, float scale, hipStream_t stream) {
  return vis_attn_prefill_rows(Q, K, Vt, O, work, n_work, Hq, Hkv, HD, Sq, k_tokens, vt_ld, ldo, causal, scale, 0, stream);
}

extern "C" int vis_attn_prefill_rows(const void* Q, const void* K, const void* Vt, void* O, const void* work,
                                     int n_work, int Hq, int Hkv, int HD, int Sq, int k_tokens, int vt_ld, int ldo,
                                     int causal, float scale, int q_row0, hipStream_t stream) {
  if (!Q || !K || !Vt || !O || !work || n_work <= 0 || Hq <= 0 || Hkv <= 0 || Hq % Hkv != 0 || q_row0 < 0) return VIS_ERR_ARG;
  if (HD != 128 && HD != 80) return VIS_ERR_ARG;
  if (Sq <= 0 || k_tokens <= 0 || vt_ld % 64 != 0 || ldo % 8 != 0 || ldo < Hq * HD) return VIS_ERR_ARG;
  if (((uintptr_t)Q | (uintptr_t)K | (uintptr_t)Vt | (uintptr_t)O | (uintptr_t)work) & 15) return VIS_ERR_ARG;
  AttnArgs p;
  p.Q = (const bf16_t*)Q; p.K = (const bf16_t*)K; p.Vt = (const bf16_t*)Vt; p.O = (bf16_t*)O;
  p.work = (const int4*)work;
  p.Sq = Sq; p.k_tokens = k_tokens; p.vt_ld = vt_ld; p.ldo = ldo; p.group = Hq / Hkv;
  p.scale_log2 = scale * 1.4426950408889634f;
  p.q_row0 = q_row0;
  p.ws_count = nullptr; p.ws_part = nullptr; p.n_pairs = 0;
  req_offsets_none(p.req);
  if (n_work > 65535) return VIS_ERR_ARG;
  const dim3 grid(Hq, n_work), block(256);
  // head_dim 80 is built for three workgroups per CU (768 slots, <= 168 VGPRs).  VIS_ATTN_OCC=2 caps residency at two
  // per CU by asking for idle dynamic LDS (A/B runs only: three per CU measured faster on every shape tried).
  static const int occ_forced = [] { const char* e = getenv("VIS_ATTN_OCC"); return e ? atoi(e) : 0; }();
  const size_t pad_lds = (HD == 80 && occ_forced == 2) ? 28 * 1024 : 0;   // 48 KiB static + 28 KiB -> two per CU
  vis_clear_error();
  if (HD == 128) {
    if (causal) hipLaunchKernelGGL((attn_prefill_kernel<128, true>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((attn_prefill_kernel<128, false>), grid, block, 0, stream, p);
  } else {
    // non-causal head_dim 80 (the ViT towers): the 32x32x16 kernel; VIS_ATTN80=16 keeps the 16x16x32 kernel (A/B runs)
    static const bool wide = [] { const char* e = getenv("VIS_ATTN80"); return !(e && atoi(e) == 16); }();
    if (causal) hipLaunchKernelGGL((attn_prefill_kernel<80, true>), grid, block, pad_lds, stream, p);
    else if (wide) hipLaunchKernelGGL(attn_vit32_kernel, grid, block, 0, stream, p);
    else hipLaunchKernelGGL((attn_prefill_kernel<80, false>), grid, block, pad_lds, stream, p);
  }
  return vis_check_launch();
}

// vis_attn_prefill_rows (head_dim 128) for the `nreq` (<= 8) requests of a prompt-pass group in ONE launch: request r reads
// Q + r * q_bs, K + kv_off[r] (host array, element offsets), Vt + r * vt_bs, work items [r * n_work, (r + 1) * n_work) of `work`
// (one list per request: the mllama cross-attention's key counts differ per image) and writes O + r * o_bs.  Per request
// bit-identical to vis_attn_prefill_rows.
extern "C" int vis_attn_prefill_rows_many(const void* Q, const void* K, const void* Vt, void* O, const void* work, int n_work,
                                          int Hq, int Hkv, int HD, int Sq, int k_tokens, int vt_ld, int ldo, int causal,
                                          float scale, int q_row0, int nreq, long long q_bs, long long vt_bs, long long o_bs,
                                          const long long* kv_off, hipStream_t stream) {
  if (!Q || !K || !Vt || !O || !work || n_work <= 0 || Hq <= 0 || Hkv <= 0 || Hq % Hkv != 0 || q_row0 < 0) return VIS_ERR_ARG;
  if (HD != 128 || nreq < 1 || nreq > VIS_MAX_REQ || !kv_off) return VIS_ERR_ARG;
  if (Sq <= 0 || k_tokens <= 0 || vt_ld % 64 != 0 || ldo % 8 != 0 || ldo < Hq * HD || n_work > 65535) return VIS_ERR_ARG;
  if (((uintptr_t)Q | (uintptr_t)K | (uintptr_t)Vt | (uintptr_t)O | (uintptr_t)work) & 15) return VIS_ERR_ARG;
  if (q_bs < 0 || vt_bs < 0 || o_bs < 0 || (q_bs | vt_bs | o_bs) % 8) return VIS_ERR_ARG;
  AttnArgs p;
  p.Q = (const bf16_t*)Q; p.K = (const bf16_t*)K; p.Vt = (const bf16_t*)Vt; p.O = (bf16_t*)O;
  p.work = (const int4*)work;
  p.Sq = Sq; p.k_tokens = k_tokens; p.vt_ld = vt_ld; p.ldo = ldo; p.group = Hq / Hkv;
  p.scale_log2 = scale * 1.4426950408889634f;
  p.q_row0 = q_row0;
  p.ws_count = nullptr; p.ws_part = nullptr; p.n_pairs = 0;
  req_offsets_none(p.req);
  for (int r = 0; r < nreq; ++r) {
    if (kv_off[r] < 0 || kv_off[r] % 8) return VIS_ERR_ARG;
    p.req.kv[r] = kv_off[r];
  }
  p.req.q_bs = q_bs; p.req.vt_bs = vt_bs; p.req.o_bs = o_bs; p.req.work_bs = n_work;
  const dim3 grid(Hq, n_work, nreq), block(256);
  vis_clear_error();
  if (causal) hipLaunchKernelGGL((attn_prefill_kernel<128, true>), grid, block, 0, stream, p);
  else hipLaunchKernelGGL((attn_prefill_kernel<128, false>), grid, block, 0, stream, p);
  return vis_check_launch();
}

// Key-split ViT attention (head_dim 80, non-causal).  Work items as in vis_attn_prefill, plus SPLIT items whose y field
// is qn | (1 | part << 1 | pair << 2) << 8: the two parts of pair `pair` cover the same query rows and the two halves
// of their key range (cut at a multiple of 64), and their results are merged inside the kernel (attn_vit32_kernel).
// One image at 16 heads is 2450 (32-row block x all keys) wave tasks for 1024 SIMDs - some get three, the launch lasts
// as long as those; with 9 of its 39 row blocks split every CU can hold two whole and one half workgroup
// (hip.plan_attn_items_split).  ws: vis_attn_split_ws_bytes(n_pairs, Hq) bytes, 256-byte aligned, ZERO before the first
// use (the kernel leaves the counters at zero); one launch at a time per workspace.
#define ATT_SPLIT_PART_BYTES ((size_t)2 * 128 * ATT_SPLIT_ROW * sizeof(float))
static size_t att_split_count_bytes(int n_pairs, int Hq) { return (((size_t)n_pairs * Hq * sizeof(int)) + 255) & ~(size_t)255; }

extern "C" int vis_attn_split_ws_bytes(int n_pairs, int Hq) {   // 0: bad arguments (or more than an int holds)
  if (n_pairs <= 0 || Hq <= 0) return 0;
  const size_t b = att_split_count_bytes(n_pairs, Hq) + (size_t)n_pairs * Hq * ATT_SPLIT_PART_BYTES;
  return b > 0x7fffffffu ? 0 : (int)b;
}

extern "C" int vis_attn_prefill_split(const void* Q, const void* K, const void* Vt, void* O, const void* work,
                                      int n_work, int Hq, int Hkv, int HD, int Sq, int k_tokens, int vt_ld, int ldo,
                                      float scale, int n_pairs, void* ws, long long ws_bytes, hipStream_t stream) {
  if (!Q || !K || !Vt || !O || !work || n_work <= 0 || Hq <= 0 || Hkv <= 0 || Hq % Hkv != 0) return VIS_ERR_ARG;
  const int need = vis_attn_split_ws_bytes(n_pairs, Hq);
  if (HD != 80 || n_pairs <= 0 || !ws || need == 0 || ws_bytes < need) return VIS_ERR_ARG;
  if (Sq <= 0 || k_tokens <= 0 || vt_ld % 64 != 0 || ldo % 8 != 0 || ldo < Hq * HD || n_work > 65535) return VIS_ERR_ARG;
  if (((uintptr_t)Q | (uintptr_t)K | (uintptr_t)Vt | (uintptr_t)O | (uintptr_t)work) & 15) return VIS_ERR_ARG;
  if ((uintptr_t)ws & 255) return VIS_ERR_ARG;
  AttnArgs p;
  p.Q = (const bf16_t*)Q; p.K = (const bf16_t*)K; p.Vt = (const bf16_t*)Vt; p.O = (bf16_t*)O;
  p.work = (const int4*)work;
  p.Sq = Sq; p.k_tokens = k_tokens; p.vt_ld = vt_ld; p.ldo = ldo; p.group = Hq / Hkv;
  p.scale_log2 = scale * 1.4426950408889634f;
  p.q_row0 = 0;
  p.ws_count = (int*)ws;
  p.ws_part = (float*)((char*)ws + att_split_count_bytes(n_pairs, Hq));
  p.n_pairs = n_pairs;
  vis_clear_error();
  hipLaunchKernelGGL(attn_vit32_kernel, dim3(Hq, n_work), dim3(256), 0, stream, p);
  return vis_check_launch();
}
