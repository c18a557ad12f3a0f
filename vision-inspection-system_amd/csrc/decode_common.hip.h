// Device code shared by the decode kernels (decode.hip) and the chained layer-head launch (decode_chain.hip):
// the weight-streaming GEMV building blocks and one work item of the split decode attention.
#pragma once
#include "common.hip.h"

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

enum { GV_ACT_NONE = 0, GV_ACT_SWIGLU = 3 };

struct GemvArgs {
  const bf16_t* x;
  const bf16_t* W;
  const bf16_t* bias;
  const bf16_t* R;
  const bf16_t* norm_w;  // non-null: x <- rmsnorm(x) * norm_w before the product
  void* y;
  int N, K, ldw;
  int act, out_f32;
  int outs_per_block;
  float eps;
  // vis_gemv_bf16_rows: nb input rows (<= the kernel's NB) share every weight read; element strides between rows
  int nb, ldx, ldy, ldr;
  // vis_gemv_bf16_argmax (AMAX kernel instance only): per-workgroup (value, index) maxima of the pick's first stage
  float* am_val;
  int* am_idx;
  const int* am_step;
  float am_inv_temp;
  unsigned am_seed;
};

__device__ __forceinline__ float gumbel_noise(unsigned seed, unsigned step, unsigned i) {
  unsigned long long z = ((unsigned long long)seed << 32) ^ ((unsigned long long)step * 0x9E3779B97F4A7C15ull) ^ i;
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  // 23 random bits + 0.5: every value k + 0.5 (k < 2^23) is exact in f32, so u lies in [2^-24, 1 - 2^-24] - never 0 or 1.
  // (24 bits + 0.5f rounds 16777215.5 up to 2^24, i.e. u = 1 and +inf noise once in 2^24 draws: at V = 152 064 logits
  // that is a garbage token in ~1 % of the sampled steps.)
  const float u = ((float)(z >> 41) + 0.5f) * (1.0f / 8388608.0f);
  // inner log in full precision: for u near 1 (the upper Gumbel tail, the draws that decide rare picks) -log u is
  // tiny and the fast log's absolute error would be a large relative one; the outer log has no such problem
  return -__logf(-logf(u));
}



// NB: __builtin_bit_cast(bf16x2, v[i]) on a vector ELEMENT is miscompiled by hipcc 7.2 (always
// element 0); extract the bf16 pairs with shufflevector from a whole-vector bit_cast instead.
__device__ __forceinline__ float dot8(const u32x4& w, const u32x4& x, float acc) {
  const bf16x8 wv = __builtin_bit_cast(bf16x8, w), xv = __builtin_bit_cast(bf16x8, x);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(wv, wv, 0, 1), __builtin_shufflevector(xv, xv, 0, 1), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(wv, wv, 2, 3), __builtin_shufflevector(xv, xv, 2, 3), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(wv, wv, 4, 5), __builtin_shufflevector(xv, xv, 4, 5), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(wv, wv, 6, 7), __builtin_shufflevector(xv, xv, 6, 7), acc, false);
  return acc;
}

// Work unit = "task": one row PAIR (two weight rows, or the gate/up rows of one SwiGLU output) x one
// K-segment of GV_SEG 16-byte chunks per lane.  A wave owns a contiguous range of pairs and walks its tasks
// with two register sets: the loads of task t+1 are issued before task t is consumed, so up to
// 2 x 2 x GV_SEG 16-byte loads per lane are in flight and the pipeline never drains between rows.
// Weight loads are non-temporal (each byte is read exactly once per token).
#define GV_SEG 8

struct GvBuf {
  u32x4 w0[GV_SEG], w1[GV_SEG];
};

__device__ __forceinline__ void gv_rows(const GemvArgs& p, bool swiglu, int pair, int& r0, int& r1) {
  if (swiglu) {
    r0 = ((pair >> 4) << 5) + (pair & 15);  // gate row in the 16-interleaved weight
    r1 = r0 + 16;                            // matching up row
  } else {
    r0 = 2 * pair;
    r1 = min(r0 + 1, p.N - 1);
  }
}

__device__ __forceinline__ void gv_load(GvBuf& b, const GemvArgs& p, bool swiglu, int pair, int seg, int lane,
                                        int nch) {
  int r0, r1;
  gv_rows(p, swiglu, pair, r0, r1);
  const bf16_t* w0 = p.W + (size_t)r0 * p.ldw;
  const bf16_t* w1 = p.W + (size_t)r1 * p.ldw;
#pragma unroll
  for (int u = 0; u < GV_SEG; ++u) {
    // unconditional loads (clamped address): a predicated load would make hipcc wait vmcnt(0) per branch
    const int c = min(lane + 64 * (seg * GV_SEG + u), nch - 1);
    b.w0[u] = __builtin_nontemporal_load((const u32x4*)(w0 + c * 8));
    b.w1[u] = __builtin_nontemporal_load((const u32x4*)(w1 + c * 8));
  }
}

// NB rows of x (xs[b * K ...]) against the task's two weight rows: the weight registers are used NB times
template <int NB>
__device__ __forceinline__ void gv_consume(const GvBuf& b, const bf16_t* xs, int K, int seg, int lane, int nch, float (&a0)[NB],
                                           float (&a1)[NB]) {
#pragma unroll
  for (int u = 0; u < GV_SEG; ++u) {
    const int c = lane + 64 * (seg * GV_SEG + u);
#pragma unroll
    for (int r = 0; r < NB; ++r) {
      u32x4 xv = *(const u32x4*)(xs + (size_t)r * K + min(c, nch - 1) * 8);
      if (c >= nch) xv = (u32x4){0u, 0u, 0u, 0u};  // clamped duplicate chunk contributes nothing
      a0[r] = dot8(b.w0[u], xv, a0[r]);
      a1[r] = dot8(b.w1[u], xv, a1[r]);
    }
  }
}

template <int NB>
__device__ __forceinline__ void gv_finish(const GemvArgs& p, bool swiglu, int pair, int lane, float (&a0)[NB], float (&a1)[NB]) {
#pragma unroll
  for (int r = 0; r < NB; ++r) {
    a0[r] = wave_sum(a0[r]);
    a1[r] = wave_sum(a1[r]);
  }
  if (lane != 0) return;
#pragma unroll
  for (int r = 0; r < NB; ++r) {
    if (r >= p.nb) break;
    if (swiglu) {
      ((bf16_t*)p.y)[(size_t)r * p.ldy + pair] = f2bf(silu_fast(a0[r]) * a1[r]);
      continue;
    }
    const int o = 2 * pair;
    const bool two = (o + 1 < p.N);
    float v0 = a0[r], v1 = a1[r];
    if (p.bias) { v0 += bf2f(p.bias[o]); if (two) v1 += bf2f(p.bias[o + 1]); }
    if (p.R) { v0 += bf2f(p.R[(size_t)r * p.ldr + o]); if (two) v1 += bf2f(p.R[(size_t)r * p.ldr + o + 1]); }
    if (p.out_f32) {
      ((float*)p.y)[(size_t)r * p.ldy + o] = v0;
      if (two) ((float*)p.y)[(size_t)r * p.ldy + o + 1] = v1;
    } else {
      ((bf16_t*)p.y)[(size_t)r * p.ldy + o] = f2bf(v0);
      if (two) ((bf16_t*)p.y)[(size_t)r * p.ldy + o + 1] = f2bf(v1);
    }
  }
}

// x -> rmsnorm(x) * w -> LDS (bf16, HF rounding order).  x and w are loaded ONCE (both loads issued before the
// reduction) and normalised from registers: hidden sizes <= 4096 give at most two 16-byte chunks per thread; the
// re-reading loop of the first version put a second L2 round trip into every norm-fused GEMV's prologue.
__device__ __forceinline__ void gv_stage_x_rmsnorm(const bf16_t* __restrict__ x, const bf16_t* __restrict__ nw, bf16_t* xs,
                                                   int nch, int K, float eps, int tid, int lane, int wave) {
  __shared__ float red[4];
  if (nch <= 512) {
    const int c0 = tid, c1 = tid + 256;
    const u32x4 z = (u32x4){0u, 0u, 0u, 0u};
    const u32x4 r0 = (c0 < nch) ? *(const u32x4*)(x + c0 * 8) : z, r1 = (c1 < nch) ? *(const u32x4*)(x + c1 * 8) : z;
    const u32x4 g0 = (c0 < nch) ? *(const u32x4*)(nw + c0 * 8) : z, g1 = (c1 < nch) ? *(const u32x4*)(nw + c1 * 8) : z;
    float f0[8], f1[8], w0[8], w1[8], o[8];
    unpack8(r0, f0); unpack8(r1, f1);
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) ss += f0[e] * f0[e] + f1[e] * f1[e];
    ss = wave_sum(ss);
    if (lane == 0) red[wave] = ss;
    __syncthreads();
    const float rstd = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)K + eps);
    unpack8(g0, w0); unpack8(g1, w1);
    if (c0 < nch) {
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = bf2f(f2bf(f0[e] * rstd)) * w0[e];
      *(u32x4*)(xs + c0 * 8) = pack8(o);
    }
    if (c1 < nch) {
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = bf2f(f2bf(f1[e] * rstd)) * w1[e];
      *(u32x4*)(xs + c1 * 8) = pack8(o);
    }
    return;
  }
  float ss = 0.f;
  for (int c = tid; c < nch; c += 256) {
    float f[8];
    unpack8(*(const u32x4*)(x + c * 8), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) ss += f[e] * f[e];
  }
  ss = wave_sum(ss);
  if (lane == 0) red[wave] = ss;
  __syncthreads();
  const float rstd = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)K + eps);
  for (int c = tid; c < nch; c += 256) {
    float f[8], w[8], o[8];
    unpack8(*(const u32x4*)(x + c * 8), f);
    unpack8(*(const u32x4*)(nw + c * 8), w);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = bf2f(f2bf(f[e] * rstd)) * w[e];
    *(u32x4*)(xs + c * 8) = pack8(o);
  }
}

// x -> LDS for long inputs (K > 8192), all global loads of a thread issued before the first LDS store (K = 18944:
// 10 chunks per thread; a plain load/store loop serialises ~10 L2 round trips in front of the weight stream).  Unconditional clamped
// loads (a predicated load makes hipcc drain vmcnt per branch); chunks past the end are simply not stored.
#define GV_STAGE_MAX 15  // 15 x 256 x 8 = 30720 elements >= the 60 KiB LDS limit of the launchers
__device__ __forceinline__ void gv_stage_x(const bf16_t* __restrict__ x, bf16_t* xs, int nch, int tid) {
  u32x4 v[GV_STAGE_MAX];
#pragma unroll
  for (int i = 0; i < GV_STAGE_MAX; ++i) v[i] = *(const u32x4*)(x + (size_t)min(tid + i * 256, nch - 1) * 8);
#pragma unroll
  for (int i = 0; i < GV_STAGE_MAX; ++i) {
    const int c = tid + i * 256;
    if (c < nch) *(u32x4*)(xs + c * 8) = v[i];
  }
}

// stage one row of x (optionally RMS-normalised) into LDS; called once per input row
__device__ __forceinline__ void gv_stage_row(const bf16_t* __restrict__ x, const bf16_t* __restrict__ norm_w, bf16_t* xs, int nch,
                                             int K, float eps, int tid, int lane, int wave) {
  if (norm_w) {
    gv_stage_x_rmsnorm(x, norm_w, xs, nch, K, eps, tid, lane, wave);
  } else if (nch > 1024) {
    gv_stage_x(x, xs, nch, tid);
  } else if (nch <= 512) {   // both loads in flight before the first store (one L2 round trip, not two)
    const u32x4 r0 = *(const u32x4*)(x + min(tid, nch - 1) * 8), r1 = *(const u32x4*)(x + min(tid + 256, nch - 1) * 8);
    if (tid < nch) *(u32x4*)(xs + tid * 8) = r0;
    if (tid + 256 < nch) *(u32x4*)(xs + (tid + 256) * 8) = r1;
  } else {
    for (int c = tid; c < nch; c += 256) *(u32x4*)(xs + c * 8) = *(const u32x4*)(x + c * 8);
  }
}


// ---------------------------------------------------------------------------
// Hand-offs between workgroups of ONE launch (the chained layer head, decode_chain.hip).  Per-CU L1s are never refreshed
// by other CUs' stores and the eight XCD L2s are not coherent with each other, so every handed-off value travels inside a
// GRANULE: one naturally aligned 8-byte word {32-bit payload, 32-bit tag} written by ONE relaxed agent-scope (`sc1`,
// write-through) store and read by `sc1` loads that go past the L1.  The tag is the launch number: a consumer polls the
// granules it needs until every tag is this launch's - the data IS the flag, so a hand-off costs one store and one load
// round trip, with no store drain, no counter and no second load after a flag (MI355X_MICROARCH.md, persistent-kernel
// price list: "handoff" rows / granules).  A first version of the chain used counters (sc1 payload -> s_waitcnt vmcnt(0)
// -> barrier -> agent-scope atomic add; consumer: poll -> barrier -> sc1 loads): ~6 us per hop, 30.5 us for the whole
// head against 27.5 us for the four launches it replaced.  Every poll loop is bounded: GV_CHAIN_SPIN_MAX polls of one
// `sc1` load round trip + s_sleep each.  Measured with the whole 860-workgroup grid polling granules that never arrive
// (tools/probes/poll_period.hip, profiles/r05_chain_poll_period.txt): 0.27 us per poll, i.e. 32 768 polls = ~9 ms - three
// orders of magnitude above the ~10 us a hand-off takes when the grid is resident.  Every 32nd poll also looks at the sync
// block's status word: once ANY wait of the launch has given up, every other wait of that launch - and, through the entry
// check of decode_chain_kernel, every later launch of the request - ends at once instead of spinning through its own bound
// (a stranded request cost minutes; now one bound, ~9 ms, then ~22 us per remaining launch).
#define GV_CHAIN_SPIN_MAX (1 << 15)
typedef unsigned long long gran_t;

__device__ __forceinline__ gran_t gr_ld(const gran_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void gr_st(gran_t* p, uint32_t payload, unsigned tag) {
  __hip_atomic_store(p, (gran_t)payload | ((gran_t)tag << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool gr_ok(gran_t v, unsigned tag) { return (unsigned)(v >> 32) == tag; }
// top of every poll iteration: a compiler barrier (the granule reads of a poll loop must be re-issued every time round -
// relaxed atomics and buffer loads alone do not forbid hoisting them) and, every 32nd poll, the "somebody gave
// up" check; true = stop waiting (workgroup-uniform only where the caller makes it so: callers treat it like a timeout)
__device__ __forceinline__ bool gr_poll_abort(const int* status, int it) {
  asm volatile("" ::: "memory");
  return (it & 31) == 31 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
}


// ---------------------------------------------------------------------------
// Fused decode attention (K4 + K11): one launch per layer does, for the single new token,
//   * M-RoPE of q and k from the packed projection row (cos/sin row = *step_ptr),
//   * the KV-cache append at slot *step_ptr (done by the split that owns that slot),
//   * GQA attention over the *step_ptr + 1 cached keys, split over the context.
// grid (Hkv, nsplit), 256 threads.  Every K and V row a thread needs is requested up front
// (<= DA_ITERS 16-byte loads each), so a block exposes one HBM latency, not one per key.
// The position lives in DEVICE memory: the launch is hipGraph-replayable.
// Partials (unnormalised o, running max in the log2 domain, sum) are merged by
// decode_attn_combine_kernel.
#define DA_ITERS 4                  // 16 keys per block iteration
#define DA_MAXKEYS (16 * DA_ITERS)  // keys per split

struct DecAttnArgs {
  const bf16_t* qkv;      // [(Hq + 2 Hkv) * 128] packed projection row of the new token (bias added)
  const float* cos_t;     // [cache_tokens][128]
  const float* sin_t;
  bf16_t* k_cache;        // [Hkv][cache_tokens][128]
  bf16_t* v_cache;
  const int* step_ptr;    // slot of the new token
  float* part_o;          // [Hq][nsplit][128]
  float* part_ml;         // [Hq][nsplit][2]
  int Hq, Hkv, cache_tokens, nsplit;
  float scale_log2;
  // batch (blockIdx.z = sequence): element strides between sequences
  long long qkv_bs, cache_bs, tab_bs;
  // cross-attention mode (mllama, TF:models/mllama/modeling_mllama.py:384-466): `qkv` holds only the Hq query heads,
  // the cache is a static set of keys/values (no rope, no append), *step_ptr = number of keys - 1, and the
  // per-head RMSNorm of q (q_norm) is applied while q is staged.
  const bf16_t* q_norm_w;  // [128] or null (self-attention mode)
  float q_eps;
  // batched streaming form only: keys [0, shared_len) (a multiple of 16) are the same in every sequence of the batch (the
  // text prefix a batch inspection shares, copied into every slot by the prompt passes) and are read from sequence 0's
  // copy - same values, so the same result bit for bit, but one HBM read + L2 hits instead of one HBM read per sequence
  int shared_len;
  // vis_decode_attn_parts: `qkv` is null and the projection row is finalised HERE from the split-K partials of the batched
  // qkv projection (vis_gemm_decode_*): column n of sequence b = bf16(sum_k part[k * part_stride + b * part_n + n] (* sx[b] *
  // sw[n]) + bias[n]) - skinny_finalize_kernel's arithmetic bit for bit (fin_plain_value), minus its launch
  const float* qkv_part;
  long long part_stride;   // floats between two partial slabs
  int part_ks, part_n;
  const bf16_t* qkv_bias;  // [part_n] or null
  const float* part_sx;    // [batch] or null (fp8 partials: raw sums of e4m3 products)
  const float* part_sw;    // [part_n]
};

static inline void da_no_parts(DecAttnArgs& p) {
  p.qkv_part = nullptr; p.part_stride = 0; p.part_ks = 0; p.part_n = 0; p.qkv_bias = nullptr; p.part_sx = nullptr; p.part_sw = nullptr;
}

// the bf16 values of columns `col0` and `col1` of sequence `seq`'s projection row (p.qkv already points at the sequence's row).
// Partial mode: every slab's word is requested before the first add (ONE memory round trip, like the bf16 read it replaces - a
// load -> add -> load chain over ksplit slabs cost more than the finalisation launch it removes); slabs past ksplit are read
// at a clamped index and dropped by a select, so the sum keeps skinny_finalize_kernel's order exactly.
#define DA_PART_UNROLL 8
__device__ __forceinline__ void da_qkv2(const DecAttnArgs& p, int seq, int col0, int col1, bf16_t& v0, bf16_t& v1) {
  if (!p.qkv_part) { v0 = p.qkv[col0]; v1 = p.qkv[col1]; return; }
  const float* b0 = p.qkv_part + (size_t)seq * p.part_n + col0;
  const float* b1 = p.qkv_part + (size_t)seq * p.part_n + col1;
  float t0[DA_PART_UNROLL], t1[DA_PART_UNROLL];
#pragma unroll
  for (int k = 0; k < DA_PART_UNROLL; ++k) {
    const size_t o = (size_t)min(k, p.part_ks - 1) * p.part_stride;
    t0[k] = b0[o];
    t1[k] = b1[o];
  }
  const bool scaled = p.part_sx != nullptr, has_b = p.qkv_bias != nullptr;
  const float sxb = scaled ? p.part_sx[seq] : 0.f;
  const float sw0 = scaled ? p.part_sw[col0] : 0.f, sw1 = scaled ? p.part_sw[col1] : 0.f;
  const float bi0 = has_b ? bf2f(p.qkv_bias[col0]) : 0.f, bi1 = has_b ? bf2f(p.qkv_bias[col1]) : 0.f;
  float a0 = t0[0], a1 = t1[0];
#pragma unroll
  for (int k = 1; k < DA_PART_UNROLL; ++k) {
    a0 = (k < p.part_ks) ? a0 + t0[k] : a0;
    a1 = (k < p.part_ks) ? a1 + t1[k] : a1;
  }
  for (int k = DA_PART_UNROLL; k < p.part_ks; ++k) { a0 += b0[(size_t)k * p.part_stride]; a1 += b1[(size_t)k * p.part_stride]; }
  v0 = f2bf(fin_plain_value(a0, scaled, sxb, sw0, has_b, bi0));
  v1 = f2bf(fin_plain_value(a1, scaled, sxb, sw1, has_b, bi1));
}
__device__ __forceinline__ bf16_t da_qkv(const DecAttnArgs& p, int seq, int col) {
  bf16_t v0, v1;
  da_qkv2(p, seq, col, col, v0, v1);
  return v0;
}

// Structure (no cross-lane reductions inside a wave):
//   scores : S^T[key][head] = K * Q^T on the MFMA (v_mfma_f32_16x16x32_bf16): a K row group of 16 keys x 128
//            dims is loaded straight from the cache into the A-operand layout (lane = key, 16 B of d), Q^T
//            (heads padded to 16) is the B operand; the accumulator already has (key, head) per lane.
//   softmax: statistics per head over the <= 128 keys of the split (LDS).
//   P * V  : each lane owns two output dims for all G heads and walks its wave's keys; V rows are read as
//            256-byte coalesced rows, p[head][key] is an LDS broadcast; waves are merged through LDS.
// LDS slot of key k of a split inside a head's score row: the keys one wave owns in P * V (k = wave + 4 i) are contiguous
__device__ __forceinline__ int da_pos(int k) { return (k & 3) * (DA_MAXKEYS / 4) + (k >> 2); }

template <int G>
struct DecAttnLds {
  __attribute__((aligned(16))) bf16_t q_s[16][128];   // heads >= G are zero
  __attribute__((aligned(16))) bf16_t knew_s[128];
  __attribute__((aligned(16))) bf16_t vnew_s[128];
  float sc[G][DA_MAXKEYS];
  float red[4][G][128];
  float ml[G][2];
  __attribute__((aligned(16))) bf16_t raw[G + 2][128];   // chained launch only: the kv group's q heads, k and v rows as handed over
};

// One (kv head, split, sequence) work item of the split decode attention.  CHAIN = false: the stand-alone launch
// (decode_attn_fused_kernel).  CHAIN = true: the same arithmetic inside the chained layer-head launch (decode_chain.hip):
// the packed projection row is produced by OTHER workgroups of the same launch, so the item first collects its kv group's
// rows from their granules (`cc`) and leaves its partials as granules for the merging workgroups.
// Returns the number of keys of the split (0: the split lies past the context and nothing was written).
struct ChainCtx {
  const gran_t* qkv_g;  // [Nqkv / 2] granule i = projection rows 2 i, 2 i + 1 (two bf16)
  gran_t* part_g;       // [Hq][nsplit][130] granules: 128 unnormalised outputs (f32 bits), running max, sum
  gran_t* cue_g;        // [Hkv] "this kv group's projection rows are all out" (published by the group's split 0)
  unsigned tag;         // this launch
  int* status;          // set to 1 when a bounded wait gave up
#ifdef CHAIN_PROBE
  unsigned long long* probe;   // this workgroup's 8 clock stamps (tools/probes/chain_probe.sh builds only)
#define CC_STAMP(i) do { if (CHAIN && threadIdx.x == 0 && cc.probe) cc.probe[i] = wall_clock64(); } while (0)
#else
#define CC_STAMP(i) do { } while (0)
#endif
};

// collect the (G + 2) x 128 bf16 values of kv group `hkv` (its G query heads, its key head, its value head) into L.raw
template <int G>
__device__ __forceinline__ void chain_stage_qkv(const ChainCtx& cc, DecAttnLds<G>& L, int hkv, int Hq, int Hkv, int tid,
                                                bool publish) {
  constexpr int N = (G + 2) * 64, NIT = (N + 255) / 256;
  const gran_t* src[NIT];
  bool pend[NIT];
#pragma unroll
  for (int k = 0; k < NIT; ++k) {
    const int i = min(tid + 256 * k, N - 1), g = i >> 6, j = i & 63;
    const int head = (g < G) ? hkv * G + g : ((g == G) ? Hq + hkv : Hq + Hkv + hkv);
    src[k] = cc.qkv_g + head * 64 + j;
    pend[k] = tid + 256 * k < N;
  }
  bool left = true;
  for (int it = 0; it < GV_CHAIN_SPIN_MAX && left; ++it) {
    if (gr_poll_abort(cc.status, it)) break;
    gran_t v[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) v[k] = gr_ld(src[k]);
    left = false;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      if (!pend[k]) continue;
      if (gr_ok(v[k], cc.tag)) { ((uint32_t*)L.raw)[tid + 256 * k] = (uint32_t)v[k]; pend[k] = false; }
      else left = true;
    }
    if (left) __builtin_amdgcn_s_sleep(2);
  }
  if (left) __hip_atomic_store(cc.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  // the projection workgroups hold their W_o loads back until the qkv weights are through (else the early finishers' W_o
  // requests queue in front of the late qkv rows: measured tail of the qkv phase 15 us instead of 9)
  if (publish && tid == 0) gr_st(cc.cue_g + hkv, 1u, cc.tag);
}

template <int G, bool CHAIN>
__device__ __forceinline__ int decode_attn_split_body(DecAttnArgs p, DecAttnLds<G>& L, int hkv, int split, int seq,
                                                      const ChainCtx& cc) {
  constexpr int HD = 128, HALF = 64;
  constexpr int KGRP = DA_MAXKEYS / 16 / 4;  // 16-key groups per wave (2)
  constexpr int VROWS = DA_MAXKEYS / 4;      // V rows per wave (32)
  auto& q_s = L.q_s; auto& knew_s = L.knew_s; auto& vnew_s = L.vnew_s; auto& sc = L.sc; auto& red = L.red; auto& ml = L.ml;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, h = lane >> 4;
  p.qkv += seq * p.qkv_bs;
  p.k_cache += seq * p.cache_bs;
  p.v_cache += seq * p.cache_bs;
  p.cos_t += seq * p.tab_bs;
  p.sin_t += seq * p.tab_bs;
  p.step_ptr += seq;
  p.part_o += (size_t)seq * p.Hq * p.nsplit * 128;
  p.part_ml += (size_t)seq * p.Hq * p.nsplit * 2;
  // Split s always owns keys [128 s, 128 s + 128): the row addresses do not depend on the position, so the
  // cache reads below are issued BEFORE *step_ptr has even arrived; splits past the context just exit.
  const int ks = split * DA_MAXKEYS;
  bf16_t* Kh = p.k_cache + (size_t)hkv * p.cache_tokens * HD;
  bf16_t* Vh = p.v_cache + (size_t)hkv * p.cache_tokens * HD;

  // ---- issue every cache read up front (unconditional, clamped rows: one exposed HBM latency per block)
  u32x4 kreg[KGRP][4];
#pragma unroll
  for (int gi = 0; gi < KGRP; ++gi) {
    const int row = min(ks + (wave + 4 * gi) * 16 + l15, p.cache_tokens - 1);
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) kreg[gi][ds] = *(const u32x4*)(Kh + (size_t)row * HD + ds * 32 + 8 * h);
  }
  uint32_t vreg[VROWS];
#pragma unroll
  for (int i = 0; i < VROWS; ++i) {
    const int row = min(ks + wave + 4 * i, p.cache_tokens - 1);
    vreg[i] = *(const uint32_t*)(Vh + (size_t)row * HD + 2 * lane);
  }

  const int slot = min(*p.step_ptr, p.cache_tokens - 1);
  const int ctx = slot + 1;
  if (ks >= ctx) return 0;  // whole block: nothing to attend to (its partials are never read by the combine)
  const int ke = min(ks + DA_MAXKEYS, ctx);
  const int nk = ke - ks;

  const bool cross = !CHAIN && p.q_norm_w != nullptr;  // workgroup-uniform
  bool owner = false;
  CC_STAMP(1);
  // the rope row of this position: requested before the projection rows are awaited (chained launch: they arrive later)
  constexpr int NROT = ((G + 1) * HALF + 255) / 256;
  float rc0[NROT], rs0[NROT], rc1[NROT], rs1[NROT];
  if (!cross) {
    const float* cr = p.cos_t + (size_t)slot * HD;
    const float* sr = p.sin_t + (size_t)slot * HD;
#pragma unroll
    for (int k = 0; k < NROT; ++k) {
      const int d = (tid + 256 * k) & (HALF - 1);
      rc0[k] = cr[d]; rs0[k] = sr[d]; rc1[k] = cr[HALF + d]; rs1[k] = sr[HALF + d];
    }
  }
  if constexpr (CHAIN) chain_stage_qkv<G>(cc, L, hkv, p.Hq, p.Hkv, tid, split == 0);   // the group's q / k / v rows, as they arrive
  CC_STAMP(2);
  if (!cross) {
    // ---- rotate q (G heads) and the new k; stage them as bf16 (exactly what later steps read back)
#pragma unroll
    for (int k = 0; k < NROT; ++k) {
      const int it = tid + 256 * k;
      if (it >= (G + 1) * HALF) break;
      const int g = it / HALF, d = it - g * HALF;
      const int head = (g < G) ? hkv * G + g : p.Hq + hkv;
      bf16_t qa, qb;
      if constexpr (CHAIN) { qa = L.raw[g][d]; qb = L.raw[g][HALF + d]; }
      else da_qkv2(p, seq, head * HD + d, head * HD + HALF + d, qa, qb);
      const float a = bf2f(qa), b = bf2f(qb);
      const bf16_t oa = f2bf(a * rc0[k] - b * rs0[k]);
      const bf16_t ob = f2bf(b * rc1[k] + a * rs1[k]);
      bf16_t* dst = (g < G) ? q_s[g] : knew_s;
      dst[d] = oa;
      dst[HALF + d] = ob;
    }
    for (int it = tid; it < (16 - G) * HD; it += 256) q_s[G + it / HD][it % HD] = 0;
    if (tid < HD) vnew_s[tid] = CHAIN ? L.raw[G + 1][tid] : da_qkv(p, seq, (p.Hq + p.Hkv + hkv) * HD + tid);
    __syncthreads();
    owner = (slot >= ks) && (slot < ke);
    if (owner && tid < HD) {  // KV-cache append (no other block reads this row in this launch)
      Kh[(size_t)slot * HD + tid] = knew_s[tid];
      Vh[(size_t)slot * HD + tid] = vnew_s[tid];
    }
  } else {
    // ---- cross-attention: q_norm (RMSNorm over the 128 dims of each head, HF rounding: normalised value to
    //      bf16, then times the weight), one wave per head
    for (int g = wave; g < G; g += 4) {
      const bf16_t* qh = p.qkv + (size_t)(hkv * G + g) * HD;
      const float a = bf2f(qh[lane]), b = bf2f(qh[lane + 64]);
      const float ss = wave_sum(a * a + b * b);
      const float rstd = rsqrtf(ss * (1.0f / HD) + p.q_eps);
      q_s[g][lane] = f2bf(bf2f(f2bf(a * rstd)) * bf2f(p.q_norm_w[lane]));
      q_s[g][lane + 64] = f2bf(bf2f(f2bf(b * rstd)) * bf2f(p.q_norm_w[lane + 64]));
    }
    for (int it = tid; it < (16 - G) * HD; it += 256) q_s[G + it / HD][it % HD] = 0;
    if (tid < HD) { knew_s[tid] = 0; vnew_s[tid] = 0; }
    __syncthreads();
  }
  const int new_row = cross ? -1 : slot;  // the cache row whose value is still only in knew_s / vnew_s

  // ---- scores on the MFMA
  bf16x8 qf[4];
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) qf[ds] = *(const bf16x8*)(&q_s[l15][ds * 32 + 8 * h]);
#pragma unroll
  for (int gi = 0; gi < KGRP; ++gi) {
    const int kbase = (wave + 4 * gi) * 16;
    if (kbase < nk) {  // wave-uniform
      const bool is_new = (ks + kbase + l15 == new_row);
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ds = 0; ds < 4; ++ds) {
        u32x4 kv = kreg[gi][ds];
        const u32x4 nv = *(const u32x4*)(&knew_s[ds * 32 + 8 * h]);
        if (is_new) kv = nv;
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kv), qf[ds], acc, 0, 0, 0);
      }
      // acc[r] = S^T[key = kbase + 4h + r][head = l15]
      if (l15 < G) {
#pragma unroll
        for (int r = 0; r < 4; ++r) sc[l15][da_pos(kbase + 4 * h + r)] = acc[r] * p.scale_log2;
      }
    }
  }
  __syncthreads();
  CC_STAMP(6);

  // ---- per-head softmax statistics over this split (one wave per head)
  for (int g = wave; g < G; g += 4) {
    float mx = -1.0e30f;
    for (int i = lane; i < nk; i += 64) mx = fmaxf(mx, sc[g][da_pos(i)]);
    mx = wave_max(mx);
    float ls = 0.f;
    for (int i = lane; i < DA_MAXKEYS; i += 64) {      // slots past the split's end get p = 0: P * V runs branch-free
      const float e = (i < nk) ? exp2f(sc[g][da_pos(i)] - mx) : 0.f;
      sc[g][da_pos(i)] = e;
      if (i < nk) ls += e;
    }
    ls = wave_sum(ls);
    if (lane == 0) { ml[g][0] = mx; ml[g][1] = ls; }
  }
  __syncthreads();
  CC_STAMP(7);

  // ---- O[g][d] += p[g][key] * V[key][d]; lane owns d = 2*lane, 2*lane+1
  float acc0[G], acc1[G];
#pragma unroll
  for (int g = 0; g < G; ++g) { acc0[g] = 0.f; acc1[g] = 0.f; }
  const uint32_t vnew = *(const uint32_t*)(&vnew_s[2 * lane]);
  // this wave's keys (wave + 4 i) sit at 16 consecutive floats of a head's row (da_pos): four ds_read_b128 per head instead
  // of 16 dependent broadcast reads (the one-read-per-key form took 3.5 us of the split's 7: tools/probes/chain_probe.py);
  // every accumulator still adds its keys in ascending order - same bits
  float v0[VROWS], v1[VROWS];
#pragma unroll
  for (int i = 0; i < VROWS; ++i) {
    uint32_t raw = (ks + wave + 4 * i == new_row) ? vnew : vreg[i];
    if (wave + 4 * i >= nk) raw = 0u;      // (a clamped duplicate row: whatever it holds, 0 x 0 adds nothing)
    v0[i] = __uint_as_float(raw << 16);
    v1[i] = __uint_as_float(raw & 0xffff0000u);
  }
#pragma unroll
  for (int g = 0; g < G; ++g) {
    f32x4 pq[VROWS / 4];
#pragma unroll
    for (int q = 0; q < VROWS / 4; ++q) pq[q] = *(const f32x4*)(&sc[g][wave * VROWS + 4 * q]);
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int i = 0; i < VROWS; ++i) {      // keys past the split's end: p = 0 and v = 0, fma(0, 0, a) = a exactly
      const float pw = pq[i >> 2][i & 3];
      a0 += pw * v0[i];
      a1 += pw * v1[i];
    }
    acc0[g] = a0;
    acc1[g] = a1;
  }
#pragma unroll
  for (int g = 0; g < G; ++g) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    *(f32x2_*)(&red[wave][g][2 * lane]) = (f32x2_){acc0[g], acc1[g]};
  }
  __syncthreads();
  CC_STAMP(3);
  for (int i = tid; i < G * HD; i += 256) {
    const int g = i / HD, d = i - g * HD;
    const float v = red[0][g][d] + red[1][g][d] + red[2][g][d] + red[3][g][d];
    const int hq = hkv * G + g;
    if constexpr (CHAIN) {
      gran_t* pg = cc.part_g + ((size_t)hq * p.nsplit + split) * 130;
      gr_st(pg + d, __float_as_uint(v), cc.tag);
      if (d == 0) {
        gr_st(pg + 128, __float_as_uint((nk > 0) ? ml[g][0] : -1.0e30f), cc.tag);
        gr_st(pg + 129, __float_as_uint((nk > 0) ? ml[g][1] : 0.f), cc.tag);
      }
    } else {
      p.part_o[((size_t)hq * p.nsplit + split) * HD + d] = v;
      if (d == 0) {
        p.part_ml[((size_t)hq * p.nsplit + split) * 2 + 0] = (nk > 0) ? ml[g][0] : -1.0e30f;
        p.part_ml[((size_t)hq * p.nsplit + split) * 2 + 1] = (nk > 0) ? ml[g][1] : 0.f;
      }
    }
  }
  return nk;
}

