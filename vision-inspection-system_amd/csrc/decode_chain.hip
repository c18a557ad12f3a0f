// The head of a decoder layer for ONE new token as ONE launch (single-sequence decode, K10 + K4 + K11 + K10):
//
//     qkv  = W_qkv rmsnorm(x) + b        (vis_gemv_bf16 with the fused norm)
//     attn = attention(rope(q), cache + rope(k), v)   (vis_decode_attn: split kernel + combine kernel)
//     y    = x + W_o attn                (vis_gemv_bf16 with the residual)
//
// Four launches in the unchained step (9.2 + 7.0 + 4.9 + 6.5 us at Qwen2-VL-7B shapes, profiles/r03_gemv_by_shape.csv)
// whose 59 MB of weights are worth 9 us of HBM time: each of them is a ramp and a drain, and the two attention launches
// move 5 MB.  Here the dependent stages are workgroup ROLES of one grid, handing over through tagged granules
// (decode_common.hip.h): the data is the flag.
//
//   blocks [0, n_gv)        "projection" workgroups, 4 waves, one weight-row pair per wave (the GEMV kernel's arithmetic,
//                            lane for lane): qkv row pair -> one granule; then they fetch their W_o row pair into the same
//                            registers - those loads fly while the attention runs -, collect the merged attention row
//                            from its granules and finish y.
//   blocks [n_gv, + Hkv * nsplit)   "attention" workgroups = the split kernel's (kv head, split) items: their K / V rows
//                            are requested at once (addresses do not depend on the projection), then they collect their
//                            kv group's q / k / v rows and leave partials (granules again).
//   blocks [.., + Hq)        "merge" workgroups, one per query head = the combine kernel (same arithmetic, same summation
//                            order) on partials collected as they arrive; they publish the head's 128 outputs.
//
// Results are bit-identical to the four-launch form (tests/test_kernels_gpu.py::test_decode_chain_equals_unchained).
// Waiting is safe because the whole grid is resident: the launcher refuses shapes whose grid exceeds what the device holds
// at this kernel's register / LDS footprint (computed from hipFuncGetAttributes, not the occupancy API), every poll loop is
// bounded (a loop that gives up raises the status word; the host checks it after the token D2H).
// The tag of a launch is (launches completed on the sync block) + 1, so nothing is ever reset between launches.
#include "decode_common.hip.h"

// sync block (ints): [0] launches completed on this block (the granule tag of a launch is this + 1), [32] status
#ifndef CH_CUE_ALL
#define CH_CUE_ALL 0       // 1: the o projection waits for every head's cue before its first full read of the row (A/B)
#endif
#define CH_EPOCH 0
#define CH_STATUS 32
#define CH_INTS 64

struct ChainArgs {
  const bf16_t* x;        // [K] layer input (also the residual of the o projection); with x_idx: the table's row 0
  const int* x_idx;       // null, or a device int: the layer input is row *x_idx of x (embedding lookup of the new token)
  int x_rows;
  const bf16_t* Wqkv;     // [Nqkv][ldw_qkv]
  const bf16_t* bqkv;     // [Nqkv] or null
  const bf16_t* norm_w;   // [K] input RMSNorm weight
  const bf16_t* Wo;       // [No][ldw_o]
  bf16_t* y;              // [No] = x + W_o attn
  int Nqkv, K, ldw_qkv, No, Ko, ldw_o;
  float eps;
  DecAttnArgs att;        // (qkv / part_o / part_ml unused: those travel as granules)
  gran_t* qkv_g;          // [Nqkv / 2]
  gran_t* attn_g;         // [Hq * 64]   granule = two bf16 of the merged attention row
  gran_t* part_g;         // [Hq][nsplit][130]
  gran_t* cue_g;          // [16] per kv group: its projection rows are all out
  int* sync;
  int n_gv, n_att;
  unsigned long long* probe;   // CHAIN_PROBE builds only
};

// -DCHAIN_PROBE (tools/probes/chain_probe.sh, never the product build): every workgroup stamps the 100 MHz clock at its
// phase boundaries into probe[blockIdx.x][8]
#ifdef CHAIN_PROBE
#define CH_STAMP(i) do { if (threadIdx.x == 0 && p.probe) p.probe[(size_t)blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#else
#define CH_STAMP(i) do { } while (0)
#endif

struct ChainMergeLds {
  float wgt[256];
  float osum[2][128];
  float wm[4];
  float outv[128];
};

// One query head's partials -> its 128 attention outputs: decode_attn_combine_kernel's arithmetic and summation order,
// operands collected from granules as they arrive.  256 threads = 128 dims x 2 split-halves.
__device__ __forceinline__ void chain_merge_head(const gran_t* part_g, gran_t* attn_g, int hq, int nsplit, int active,
                                                 unsigned tag, int* status, ChainMergeLds& M_, int tid,
                                                 unsigned long long* probe) {
#ifdef CHAIN_PROBE
#define MG_STAMP(i) do { if (tid == 0 && probe) probe[(size_t)blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#else
#define MG_STAMP(i) do { } while (0)
#endif
  constexpr int HD = 128, PRE = 32;                 // PRE x 2 split slots preloaded per thread (4096 cached keys)
  const int d = tid & 127, half = tid >> 7;
  const gran_t* pg = part_g + (size_t)hq * nsplit * 130;
  bool bad = false;
  // ---- everything this thread needs is requested at once: its split-half's partial rows (they do not depend on the
  //      statistics) and the statistics of split `tid`; re-read until every tag is this launch's
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)pg, 0, nsplit * 1040, 0x00020000);
  gran_t gv[PRE];
  float m = -1.0e30f, l = 0.f;
  if (tid < 64) {                       // first sign of life (split 0's statistics): until then one granule per poll
    for (int it = 0; it < GV_CHAIN_SPIN_MAX; ++it) {
      if (gr_poll_abort(status, it)) break;
      if (__all(gr_ok(gr_ld(pg + 128), tag))) break;
      __builtin_amdgcn_s_sleep(4);
    }
  }
  __syncthreads();
  {
    bool ok = false;
    const int st = min(tid, max(active - 1, 0));
    for (int it = 0; it < GV_CHAIN_SPIN_MAX && !ok; ++it) {
      if (gr_poll_abort(status, it)) break;
#pragma unroll
      for (int j = 0; j < PRE; ++j) {     // wave-uniform row (scalar offset) + the lane's dim: no per-load address registers
        const int sj = __builtin_amdgcn_readfirstlane(min(half + 2 * j, max(active - 1, 0)));
        const u32x2 r = __builtin_amdgcn_raw_buffer_load_b64(rs, (unsigned)d * 8u, (unsigned)sj * 1040u, 16 /* sc1 */);
        gv[j] = (gran_t)r[0] | ((gran_t)r[1] << 32);
      }
      const gran_t gm = gr_ld(pg + (size_t)st * 130 + 128), gl = gr_ld(pg + (size_t)st * 130 + 129);
      ok = true;
#pragma unroll
      for (int j = 0; j < PRE; ++j)
        if (half + 2 * j < active && !gr_ok(gv[j], tag)) ok = false;
      if (tid < active) {
        if (gr_ok(gm, tag) && gr_ok(gl, tag)) { m = __uint_as_float((uint32_t)gm); l = __uint_as_float((uint32_t)gl); }
        else ok = false;
      }
      if (!ok) __builtin_amdgcn_s_sleep(2);
    }
    bad |= !ok;
  }
  MG_STAMP(1);
  float M = wave_max(m);
  if ((tid & 63) == 0) M_.wm[tid >> 6] = M;
  __syncthreads();
  M = fmaxf(fmaxf(M_.wm[0], M_.wm[1]), fmaxf(M_.wm[2], M_.wm[3]));
  const float w = (tid < active) ? exp2f(m - M) : 0.f;
  M_.wgt[tid] = w;
  const float lw = wave_sum(w * l);
  __syncthreads();
  if ((tid & 63) == 0) M_.wm[tid >> 6] = lw;
  MG_STAMP(2);
  float o = 0.f;
#pragma unroll
  for (int j = 0; j < PRE; ++j) {
    const int s = half + 2 * j;
    if (s < active) o += M_.wgt[s] * __uint_as_float((uint32_t)gv[j]);
  }
  for (int s = half + 2 * PRE; s < active; s += 2) {   // contexts past 64 splits
    gran_t g = 0;
    bool ok = false;
    for (int it = 0; it < GV_CHAIN_SPIN_MAX; ++it) {
      if (gr_poll_abort(status, it)) break;
      g = gr_ld(pg + (size_t)s * 130 + d);
      if (gr_ok(g, tag)) { ok = true; break; }
      __builtin_amdgcn_s_sleep(2);
    }
    bad |= !ok;
    o += M_.wgt[s] * __uint_as_float((uint32_t)g);
  }
  M_.osum[half][d] = o;
  __syncthreads();
  if (tid < HD) {
    const float lt = M_.wm[0] + M_.wm[1] + M_.wm[2] + M_.wm[3];
    const float ot = M_.osum[0][tid] + M_.osum[1][tid];
    M_.outv[tid] = lt > 0.f ? ot / lt : 0.f;
  }
  __syncthreads();
  if (tid < HD / 2)
    gr_st(attn_g + hq * 64 + tid, (uint32_t)f2bf(M_.outv[2 * tid]) | ((uint32_t)f2bf(M_.outv[2 * tid + 1]) << 16), tag);
  MG_STAMP(3);
  if (bad) __hip_atomic_store(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int G>
__global__ __launch_bounds__(256) void decode_chain_kernel(ChainArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x;
  const unsigned e = (unsigned)__hip_atomic_load(p.sync + CH_EPOCH, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // tag 0 is what a zero-initialised (never written) granule carries: the launch after epoch 2^32 - 1 takes tag 1 (the
  // engine zeroes workspace and sync block long before a counter gets there: Qwen2VLEngine._chain_epoch_guard)
  const unsigned tag = (e + 1u != 0u) ? e + 1u : 1u;
  int* status = p.sync + CH_STATUS;
  const int Hkv = p.att.Hkv, Hq = p.att.Hq;
  CH_STAMP(0);
  // Fail fast: a wait of an EARLIER launch on this sync block gave up - this request's results are
  // already invalid and the host will serve it again (hip.ChainStalled).  Nobody waits for anybody: the launch is over in
  // one dispatch; workgroup 0 still publishes the launch number so that the block stays consistent until the host zeroes it.
  if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
    if (b == 0 && tid == 0) __hip_atomic_store(p.sync + CH_EPOCH, (int)tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }

  if (b < p.n_gv) {
    // ------------------------------------------------------------------ projection role
    bf16_t* xs = (bf16_t*)smem;
    const int wid = b * 4 + wave;
    if (p.x_idx) p.x += (size_t)min(max(*p.x_idx, 0), p.x_rows - 1) * p.K;   // (vis_gather_rows' clamp)
    GemvArgs ga;
    ga.W = p.Wqkv; ga.ldw = p.ldw_qkv; ga.N = p.Nqkv;
    const int nch = p.K >> 3, n_pairs_q = p.Nqkv >> 1;
    GvBuf A;
    gv_load(A, ga, false, min(wid, n_pairs_q - 1), 0, lane, nch);   // in flight while x is normalised
    gv_stage_x_rmsnorm(p.x, p.norm_w, xs, nch, p.K, p.eps, tid, lane, wave);
    __syncthreads();
    CH_STAMP(1);
    float a0[1] = {0.f}, a1[1] = {0.f};
    gv_consume<1>(A, xs, p.K, 0, lane, nch, a0, a1);
    float s0 = wave_sum(a0[0]), s1 = wave_sum(a1[0]);
    if (lane == 0 && wid < n_pairs_q) {
      const int o = 2 * wid;
      if (p.bqkv) { s0 += bf2f(p.bqkv[o]); s1 += bf2f(p.bqkv[o + 1]); }
      gr_st(p.qkv_g + wid, (uint32_t)f2bf(s0) | ((uint32_t)f2bf(s1) << 16), tag);
    }
    CH_STAMP(2);
    // ---- o projection: this wave's W_o row pair is fetched NOW and waits in registers for the attention
    const int n_pairs_o = p.No >> 1;
    if (b * 4 >= n_pairs_o) return;                    // (workgroup-uniform) no o rows here
    __syncthreads();                                   // every wave is done with xs
    if (wave == 0) {                                   // hold the W_o requests back until every kv group's rows are out
      const gran_t* cue = p.cue_g + min(lane, Hkv - 1);
      for (int it = 0; it < GV_CHAIN_SPIN_MAX; ++it) {
        if (gr_poll_abort(status, it)) break;
        if (__all(gr_ok(gr_ld(cue), tag))) break;
        __builtin_amdgcn_s_sleep(4);
      }
    }
    __syncthreads();
    CH_STAMP(6);
    GemvArgs go;
    go.W = p.Wo; go.ldw = p.ldw_o; go.N = p.No;
    const int nch_o = p.Ko >> 3;
    gv_load(A, go, false, min(wid, n_pairs_o - 1), 0, lane, nch_o);
    // cue: the last granule of every head (cheap to poll: Hq words) until the FIRST head is out - the heads finish within
    // ~2 us of each other -; from then on the whole row is polled, every granule checked
    if (wave == 0) {
      const gran_t* cue = p.attn_g + min(lane, Hq - 1) * 64 + 63;
      for (int it = 0; it < GV_CHAIN_SPIN_MAX; ++it) {
        if (gr_poll_abort(status, it)) break;
        const bool ok = gr_ok(gr_ld(cue), tag);
        if (CH_CUE_ALL ? __all(ok) : __any(ok)) break;
        __builtin_amdgcn_s_sleep(8);
      }
    }
    __syncthreads();
    CH_STAMP(3);
    {
      constexpr int NST = 8;                           // Ko / 2 <= 2048 granules on 256 threads
      const int ng = p.Ko >> 1;
      gran_t g[NST];
      bool ok = false;
      for (int it = 0; it < GV_CHAIN_SPIN_MAX && !ok; ++it) {
#pragma unroll
        for (int k = 0; k < NST; ++k) g[k] = gr_ld(p.attn_g + min(tid + 256 * k, ng - 1));
        ok = true;
#pragma unroll
        for (int k = 0; k < NST; ++k)
          if (tid + 256 * k < ng && !gr_ok(g[k], tag)) ok = false;
        if (!ok) {
          // (the "somebody gave up" check sits HERE, where the granule registers are dead: at the top of the loop it cost
          // 18 VGPRs - 141 instead of 123, i.e. three waves per SIMD instead of four and a grid that no longer fits)
          if (gr_poll_abort(status, it)) break;
          __builtin_amdgcn_s_sleep(2);
        }
      }
#pragma unroll
      for (int k = 0; k < NST; ++k)
        if (tid + 256 * k < ng) ((uint32_t*)xs)[tid + 256 * k] = (uint32_t)g[k];
      if (!ok) __hip_atomic_store(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    CH_STAMP(4);
    a0[0] = 0.f; a1[0] = 0.f;
    gv_consume<1>(A, xs, p.Ko, 0, lane, nch_o, a0, a1);
    s0 = wave_sum(a0[0]); s1 = wave_sum(a1[0]);
    if (lane == 0 && wid < n_pairs_o) {
      const int o = 2 * wid;
      s0 += bf2f(p.x[o]); s1 += bf2f(p.x[o + 1]);
      *(uint32_t*)(p.y + o) = (uint32_t)f2bf(s0) | ((uint32_t)f2bf(s1) << 16);
    }
    // the launch is over for everybody who needed its number: workgroup 0 saw every head merged, which needed every active
    // split, which needed every projection workgroup (all of them read the number before they produced)
    CH_STAMP(5);
    if (b == 0 && tid == 0) __hip_atomic_store(p.sync + CH_EPOCH, (int)tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }

  if (b < p.n_gv + p.n_att) {
    // ------------------------------------------------------------------ attention role
    const int ba = b - p.n_gv;
    const int hkv = ba % Hkv, split = ba / Hkv;
    DecAttnLds<G>& L = *(DecAttnLds<G>*)smem;
#ifdef CHAIN_PROBE
    const ChainCtx cc{p.qkv_g, p.part_g, p.cue_g, tag, status, p.probe ? p.probe + (size_t)b * 8 : nullptr};
#else
    const ChainCtx cc{p.qkv_g, p.part_g, p.cue_g, tag, status};
#endif
    decode_attn_split_body<G, true>(p.att, L, hkv, split, 0, cc);
    CH_STAMP(5);
    return;
  }

  // -------------------------------------------------------------------- merge role: one query head
  const int hq = b - p.n_gv - p.n_att;
  const int ctx = min(*p.att.step_ptr, p.att.cache_tokens - 1) + 1;
  const int active = min((ctx + DA_MAXKEYS - 1) / DA_MAXKEYS, p.att.nsplit);   // splits that run this launch
  chain_merge_head(p.part_g, p.attn_g, hq, p.att.nsplit, active, tag, status, *(ChainMergeLds*)smem, tid, p.probe);
}

template <int G>
static size_t chain_lds_bytes(int K) {
  size_t n = sizeof(DecAttnLds<G>);
  if (sizeof(ChainMergeLds) > n) n = sizeof(ChainMergeLds);
  if ((size_t)K * 2 > n) n = (size_t)K * 2;
  return (n + 15) & ~(size_t)15;
}

// workgroups of this kernel the device holds at once, from the kernel's own footprint (the occupancy API reads one block per
// CU high for some SGPR counts - MI355X_MICROARCH.md; here a short count would make a workgroup wait for one never placed).
// The model is an otherwise EMPTY device: whatever else runs there at the same time (another stream's prompt pass, another
// process) is covered by the bounded waits and the status word only - CH_RESIDENT_MARGIN workgroup slots are left unclaimed
// so that a neighbour's small kernels do not turn straight into timeouts.  Cached per (device, dynamic LDS bytes): the
// first call on a device also raises the kernel's dynamic-LDS limit THERE (a process-wide static kept the
// first device's / first K's answer).
#define CH_RESIDENT_MARGIN 32
#define CH_MAX_DEVICES 16
template <int G>
static int chain_resident_blocks(size_t lds) {
  struct Entry { size_t lds; int blocks; bool attr_set; };
  static Entry cache[CH_MAX_DEVICES] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= CH_MAX_DEVICES) return 0;
  Entry& en = cache[dev];
  if (!en.attr_set) {
    if (hipFuncSetAttribute((const void*)decode_chain_kernel<G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(64 * 1024)) !=
        hipSuccess)
      return 0;
    en.attr_set = true;
  }
  if (en.lds == lds && en.blocks > 0) return en.blocks;
  hipFuncAttributes fa;
  if (hipFuncGetAttributes(&fa, (const void*)decode_chain_kernel<G>) != hipSuccess) return 0;
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  const int vg = ((fa.numRegs > 0 ? fa.numRegs : 512) + 7) / 8 * 8;
  int per_simd = 512 / vg;                         // a 256-thread workgroup puts one wave on each of the CU's four SIMDs
  if (per_simd > 8) per_simd = 8;
  const int by_lds = (int)((160 * 1024) / (lds + fa.sharedSizeBytes + 256));
  const int per_cu = per_simd < by_lds ? per_simd : by_lds;
  en.lds = lds;
  en.blocks = per_cu * cus - CH_RESIDENT_MARGIN;
  return en.blocks;
}

static unsigned long long* g_chain_probe = nullptr;
#ifdef CHAIN_PROBE
extern "C" void vis_decode_chain_set_probe(void* p) { g_chain_probe = (unsigned long long*)p; }
#endif

extern "C" int vis_decode_chain_sync_ints(void) { return CH_INTS; }

// bytes of the granule workspace of vis_decode_chain (zero-initialised by the caller, one per sync block)
extern "C" long long vis_decode_chain_ws_bytes(int Hq, int Hkv, int nsplit) {
  if (Hq <= 0 || Hkv <= 0 || nsplit <= 0) return 0;
  return (long long)sizeof(gran_t) * ((long long)(Hq + 2 * Hkv) * 64 + (long long)Hq * 64 + (long long)Hq * nsplit * 130 + 16);
}

// One launch for qkv projection (+ RMSNorm, bias) -> rope / KV append / attention -> o projection (+ residual); see the top
// of this file.  `sync`: vis_decode_chain_sync_ints() zero-initialised ints, `ws`: vis_decode_chain_ws_bytes() zero-initialised
// bytes; both owned by the caller's stream (one pair per engine; zero both again after a non-zero status word).
// Returns VIS_ERR_UNSUPPORTED (nothing launched; the caller then uses the four launches) for shapes the chained form does not
// cover: head_dim != 128, a group size other than 1 / 2 / 4 / 7 / 8, K or Hq * 128 above 4096, more than 64 query heads, a grid
// larger than the device holds resident.  VIS_ERR_ARG is a caller bug (null / misaligned pointers, inconsistent sizes).
// ctx_bound: the caller's promise about the largest context (cached keys incl. the new one) any launch with these arguments
// will see - a captured launch is replayed at growing positions - or <= 0 for cache_tokens.  Only the workgroups that WAIT
// need to be resident together: the projection and merge roles and the attention items of splits the context reaches; an
// item past the context returns at once, whenever it is placed, so the bound counts ceil(ctx_bound / 64) splits, not
// nsplit.  (Llama-3.2-11B: 768 + 32 + 8 splits-per-64-keys leaves room for 1536 keys; with nsplit counted a 2048-row cache
// was refused outright.)  vis_decode_chain_ctx_limit gives the largest bound the device holds.
template <int G>
static int chain_waiting_limit(int Hq, int Hkv, int K) {
  const int Nqkv = (Hq + 2 * Hkv) * 128, Ko = Hq * 128;
  const size_t lds = chain_lds_bytes<G>(K > Ko ? K : Ko);
  if (lds > 64 * 1024) return 0;
  const int resident = chain_resident_blocks<G>(lds);
  const int fixed = Nqkv / 8 + Hq;
  return resident > fixed ? (resident - fixed) / Hkv : 0;       // attention splits that may be active
}

static int chain_split_limit(int Hq, int Hkv, int K) {
  if (Hq <= 0 || Hkv <= 0 || Hq > 64 || Hkv > 16 || Hq % Hkv != 0 || K <= 0 || K % 8 != 0 || K > 4096 || Hq * 128 > 4096) return 0;
  if (K / 2 > (Hq + 2 * Hkv) * 128 / 8 * 4) return 0;
  switch (Hq / Hkv) {
    case 1: return chain_waiting_limit<1>(Hq, Hkv, K);
    case 2: return chain_waiting_limit<2>(Hq, Hkv, K);
    case 4: return chain_waiting_limit<4>(Hq, Hkv, K);
    case 7: return chain_waiting_limit<7>(Hq, Hkv, K);
    case 8: return chain_waiting_limit<8>(Hq, Hkv, K);
    default: return 0;
  }
}

// largest context (cached keys) for which vis_decode_chain accepts (Hq, Hkv, head_dim 128, hidden K) on the current device;
// 0: the chained form does not cover the shape.  The engines run the chained launch while prompt + generated tokens stay
// within it and the four launches beyond (same results bit for bit).
extern "C" int vis_decode_chain_ctx_limit(int Hq, int Hkv, int K) {
  const int splits = chain_split_limit(Hq, Hkv, K);
  return splits > 0 ? splits * DA_MAXKEYS : 0;
}

extern "C" int vis_decode_chain(const void* x, const void* x_idx, int x_rows, const void* Wqkv, const void* bqkv, const void* norm_w, const void* Wo, void* y,
                                const void* cos_t, const void* sin_t, void* k_cache, void* v_cache, const void* step_ptr,
                                void* ws, void* sync, int Hq, int Hkv, int HD, int K, int ldw_qkv, int ldw_o,
                                int cache_tokens, int nsplit, int ctx_bound, float scale, float eps, hipStream_t stream) {
  if (!x || !Wqkv || !norm_w || !Wo || !y || !cos_t || !sin_t || !k_cache || !v_cache || !step_ptr || !ws || !sync)
    return VIS_ERR_ARG;
  if (Hq <= 0 || Hkv <= 0 || HD <= 0 || K <= 0) return VIS_ERR_ARG;
  if (HD != 128 || Hq > 64 || Hkv > 16 || Hq % Hkv != 0) return VIS_ERR_UNSUPPORTED;
  const int G = Hq / Hkv;
  if (G != 1 && G != 2 && G != 4 && G != 7 && G != 8) return VIS_ERR_UNSUPPORTED;
  const int Nqkv = (Hq + 2 * Hkv) * 128, Ko = Hq * 128, No = K;
  if (K % 8 != 0 || K > 4096 || Ko > 4096 || (No & 1)) return VIS_ERR_UNSUPPORTED;   // one K-segment per weight row
  if (ldw_qkv % 8 != 0 || ldw_qkv < K || ldw_o % 8 != 0 || ldw_o < Ko) return VIS_ERR_ARG;
  if (nsplit <= 0 || nsplit > 256 || cache_tokens <= 0 || (long long)nsplit * DA_MAXKEYS < cache_tokens) return VIS_ERR_ARG;
  if (((uintptr_t)x | (uintptr_t)Wqkv | (uintptr_t)norm_w | (uintptr_t)Wo | (uintptr_t)k_cache | (uintptr_t)v_cache) & 15)
    return VIS_ERR_ARG;
  if (((uintptr_t)y | (uintptr_t)sync) & 3 || ((uintptr_t)ws & 7)) return VIS_ERR_ARG;
  ChainArgs p;
  if (x_idx && x_rows <= 0) return VIS_ERR_ARG;
  p.x_idx = (const int*)x_idx; p.x_rows = x_rows;
  p.x = (const bf16_t*)x; p.Wqkv = (const bf16_t*)Wqkv; p.bqkv = (const bf16_t*)bqkv; p.norm_w = (const bf16_t*)norm_w;
  p.Wo = (const bf16_t*)Wo; p.y = (bf16_t*)y;
  p.Nqkv = Nqkv; p.K = K; p.ldw_qkv = ldw_qkv; p.No = No; p.Ko = Ko; p.ldw_o = ldw_o; p.eps = eps;
  p.att.qkv = nullptr; p.att.cos_t = (const float*)cos_t; p.att.sin_t = (const float*)sin_t;
  p.att.k_cache = (bf16_t*)k_cache; p.att.v_cache = (bf16_t*)v_cache; p.att.step_ptr = (const int*)step_ptr;
  p.att.part_o = nullptr; p.att.part_ml = nullptr;
  p.att.Hq = Hq; p.att.Hkv = Hkv; p.att.cache_tokens = cache_tokens; p.att.nsplit = nsplit;
  p.att.scale_log2 = scale * 1.4426950408889634f;
  p.att.qkv_bs = 0; p.att.cache_bs = 0; p.att.tab_bs = 0;
  p.att.q_norm_w = nullptr; p.att.q_eps = 0.f;
  p.att.shared_len = 0;
  da_no_parts(p.att);
  p.qkv_g = (gran_t*)ws;
  p.attn_g = p.qkv_g + Nqkv / 2;
  p.part_g = p.attn_g + Hq * 64;
  p.cue_g = p.part_g + (size_t)Hq * nsplit * 130;
  p.sync = (int*)sync;
  p.probe = g_chain_probe;
  p.n_gv = Nqkv / 8;                                   // four waves x one row pair
  p.n_att = Hkv * nsplit;
  if (No / 2 > p.n_gv * 4) return VIS_ERR_UNSUPPORTED; // every o row pair needs a wave
  const int grid = p.n_gv + p.n_att + Hq;
  size_t lds = 0;
  switch (G) {
    case 1: lds = chain_lds_bytes<1>(K > Ko ? K : Ko); break;
    case 2: lds = chain_lds_bytes<2>(K > Ko ? K : Ko); break;
    case 4: lds = chain_lds_bytes<4>(K > Ko ? K : Ko); break;
    case 7: lds = chain_lds_bytes<7>(K > Ko ? K : Ko); break;
    default: lds = chain_lds_bytes<8>(K > Ko ? K : Ko); break;
  }
  const int bound = (ctx_bound > 0 && ctx_bound < cache_tokens) ? ctx_bound : cache_tokens;
  int active = (bound + DA_MAXKEYS - 1) / DA_MAXKEYS;
  if (active > nsplit) active = nsplit;
  if (lds > 64 * 1024 || active > chain_split_limit(Hq, Hkv, K)) return VIS_ERR_UNSUPPORTED;
  vis_clear_error();
  switch (G) {
    case 1: hipLaunchKernelGGL(decode_chain_kernel<1>, dim3(grid), dim3(256), lds, stream, p); break;
    case 2: hipLaunchKernelGGL(decode_chain_kernel<2>, dim3(grid), dim3(256), lds, stream, p); break;
    case 4: hipLaunchKernelGGL(decode_chain_kernel<4>, dim3(grid), dim3(256), lds, stream, p); break;
    case 7: hipLaunchKernelGGL(decode_chain_kernel<7>, dim3(grid), dim3(256), lds, stream, p); break;
    default: hipLaunchKernelGGL(decode_chain_kernel<8>, dim3(grid), dim3(256), lds, stream, p); break;
  }
  return vis_check_launch();
}
