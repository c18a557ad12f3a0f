// K10 (batched decode), round 5: the decode projection of up to 64 in-flight sequences as ONE launch - weight stream, split-K
// reduction AND the row-wise epilogue (bias / residual / SwiGLU / the next RMSNorm / the next projection's e4m3 input).
//
// r02-r04 ran every projection as gemm_decode_stream_kernel (gemm_bf16.hip: persistent stream-K workgroups, f32 partial
// slabs part[slot][rows][N], unused slots zero-filled) + skinny_finalize_kernel (decode_batched.hip: one workgroup per
// sequence sums the slots and owns the row for the norm).  profiles/r04_decode_stream_traffic.json: 156 MB of HBM traffic
// per launch for 125 MB of weights (x 1.25), and the four finalisation launches of a layer were 20 us of its 180 us at 64
// sequences, 22 of 91 us at the 4 sequences of the fp8 configuration.  Here:
//
//  * the SAME stream (one persistent workgroup per CU, LDS-DMA ring, counted vmcnt, one barrier per K-step, per-segment
//    accumulator sets stored after the last step), the same K order per tile, the same fixed summation order over a tile's
//    segments - so a row's result still depends on that row alone (slot / batch-size invariance stays bit-exact);
//  * a tile touched by ONE workgroup never leaves the registers;
//  * a tile cut between workgroups: every workgroup stores its f32 segment into a compact block of its own (no zero
//    fill, 16-byte write-through stores), drains, takes a ticket on the tile's counter (one agent-scope atomic add by one
//    lane, MI355X_MICROARCH.md "Valid forms", first table row) and the workgroup whose ticket is the last one sums the
//    blocks in segment order (its own from registers) and runs the epilogue.  Nobody waits for anybody: no residency
//    assumption, no bounded spin, no status word;
//  * the RMSNorm in front of the NEXT projection needs the whole row, which no tile owner has.  It is split in two exact
//    halves instead: the producer's epilogue writes y (the residual stream), yw = bf16(y * norm_w[n]) - the next
//    projection's A operand, a per-column factor - and the tile's partial sum of squares ssq[tile][row]; the consumer
//    multiplies its finished sums by rs[row] = rsqrt(sum_tiles ssq / K + eps), a per-row factor that commutes with the
//    projection.  One bf16 rounding (of y * w) instead of HF's two (TF modeling_qwen2_vl.py:96-110 rounds y * rstd, then
//    times w): closer to the fp32 oracle, not bit-identical to the single-sequence GEMV path (which never was required);
//  * fp8 (BASELINE configs[4]): the activations a projection consumes are MX-style blocks - OCP e4m3 bytes with one E8M0
//    (power of two) scale per row and 32 consecutive columns, exactly what one lane feeds
//    v_mfma_scale_f32_16x16x128_f8f6f4 per K-step, so the scale rides in the instruction's scale operand - produced by the
//    producer's epilogue (a wave owns 32 columns of every row of its tile: block maximum = 8 lane-local values + two
//    permlane swaps).  The per-row activation scale of r02-r04 needed the whole row (hence a finalisation launch).
//    Weights keep their per-output-row f32 scale sw[n], applied in the epilogue.
#include "decode_proj_common.hip.h"

template <bool FP8, int MB>
struct DsGeom {
  static constexpr int XI = (MB == 4) ? 2 : 1;                          // x LDS-DMA instructions per thread per stage
  static constexpr int A_BYTES = XI * 32 * DS_ROWB;                     // x tile: 32 (MB = 4: 64) rows
  static constexpr int S_BYTES = FP8 ? 4 * 64 * 4 : 0;                  // FP8: scale words, one wave-private copy per wave
  static constexpr int STAGE_BYTES = A_BYTES + DS_BN * DS_ROWB + S_BYTES;
  static constexpr int PER = XI + 4 + (FP8 ? 1 : 0);                    // vm instructions per thread per stage
  static constexpr int DEPTH = (MB == 4) ? 6 : 7;
  static constexpr int LDS_BYTES = DEPTH * STAGE_BYTES;
  static constexpr int NSEG = (MB == 4) ? 6 : 8;                        // accumulator sets kept until the end of the range
  static constexpr int GRP = 16 / (MB * 2) * 2;                         // peers summed per round trip: 32 loads in flight
};

template <int PER, int DEPTH>
__device__ __forceinline__ void ds_wait_stages(int younger) {
  // wait until all but `younger` (0 .. DEPTH - 2) most recent stages of this wave have landed
  if (younger <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
  else if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
  else if (younger == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PER) : "memory");
  else if (younger == 4 || DEPTH == 6) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * PER) : "memory");
  else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * PER) : "memory");
}

template <bool FP8, int MB>
__global__ __launch_bounds__(256, 1) void decode_proj_kernel(DsArgs p) {
  using G = DsGeom<FP8, MB>;
  constexpr int XI = G::XI, DEPTH = G::DEPTH, STAGE_BYTES = G::STAGE_BYTES, A_BYTES = G::A_BYTES, NSEG = G::NSEG;
  static_assert(5 * G::PER < 64, "vmcnt is a 6-bit counter");
  extern __shared__ __attribute__((aligned(16))) char ring[];
  __shared__ int tick_s[NSEG];
  __shared__ int segt_s[NSEG];
  __shared__ float rs_s[64];
  __shared__ float red_s[4][64];
  const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
  const int l15 = lane & 15, h = lane >> 4;
  const int s0 = blockIdx.x * p.spb;
  const int nsteps = min(s0 + p.spb, p.total) - s0;
  if (nsteps <= 0) return;  // whole workgroup

  // rs[row] of the deferred RMSNorm: computed at the end of the range (finish_sets), behind the partial stores' drain
  float rs[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) rs[mb] = 1.0f;

  // ---- staging: x tile 32 (MB = 4: 64) rows x 8 chunks (rows >= M repeat row M - 1), W tile 128 x 8 (4 per thread)
  uint32_t a_off[XI], w_off[4], s_off = 0;
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int row = i * 32 + (tid >> 3), ch = (tid & 7) ^ (row & 7);
    a_off[i] = (uint32_t)min(row, p.M - 1) * (uint32_t)p.lda + ch * 16;
  }
  if (FP8) s_off = (uint32_t)min(lane, p.M - 1) * (uint32_t)p.ldas;   // this lane's row of scale bytes (4 per K-step)
  int p_tile = s0 / p.nk_all, p_kt = s0 - p_tile * p.nk_all;  // producer cursor
  const char* a_ptr;
  const char* w_ptr;
  const char* s_ptr = nullptr;
  auto enter_tile = [&](int tile, int kt) {
    const int n0 = tile * DS_BN;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = i * 256 + tid;
      const int row = c >> 3, ch = (c & 7) ^ (row & 7);
      w_off[i] = (uint32_t)(min(n0 + row, p.N - 1) - n0) * (uint32_t)p.ldw + ch * 16;
    }
    a_ptr = p.A + (size_t)kt * DS_ROWB;
    w_ptr = p.W + (size_t)n0 * p.ldw + (size_t)kt * DS_ROWB;
    if (FP8) s_ptr = (const char*)p.As + (size_t)kt * 4;
  };
  enter_tile(p_tile, p_kt);
  const int wave_base = __builtin_amdgcn_readfirstlane(tid >> 6) * 1024;

  auto stage = [&](int slot) {  // next step of this workgroup's range -> ring slot
    char* base = ring + slot * STAGE_BYTES + wave_base;
#pragma unroll
    for (int i = 0; i < XI; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_ptr + a_off[i]),
                                       (__attribute__((address_space(3))) void*)(base + i * 4096), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_ptr + w_off[i]),
                                       (__attribute__((address_space(3))) void*)(base + A_BYTES + i * 4096), 16, 0, 0);
    if (FP8)   // the K-step's four scale bytes of every row, one word per lane, wave-private copy (256 B per wave)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s_ptr + s_off),
                                       (__attribute__((address_space(3))) void*)(ring + slot * STAGE_BYTES + A_BYTES + DS_BN * DS_ROWB +
                                                                                 (wave_base >> 2)),
                                       4, 0, 0);
    a_ptr += DS_ROWB;
    w_ptr += DS_ROWB;
    if (FP8) s_ptr += 4;
    if (++p_kt == p.nk_all) {
      p_kt = 0;
      ++p_tile;
      enter_tile(p_tile, 0);  // never dereferenced past the last tile: the caller stops issuing at nsteps
    }
  };

  const int sw7 = lane & 7;
  const int rd0 = l15 * 128 + (((0 + h) ^ sw7) << 4);
  const int rd1 = l15 * 128 + (((4 + h) ^ sw7) << 4);
  const int w_rd = A_BYTES + wn * 32 * 128;

  int c_tile = s0 / p.nk_all, c_kt = s0 - c_tile * p.nk_all;  // consumer cursor
  const int pre = min(DEPTH - 1, nsteps);
  for (int s = 0; s < pre; ++s) stage(s);
  int slot = 0, fill = pre % DEPTH;  // slot consumed this step / slot refilled this step
  int st = 0;

  // one K-step of the ring into `acc`
  auto step = [&](f32x4 (&acc)[MB][2]) __attribute__((always_inline)) {
    ds_wait_stages<G::PER, DEPTH>(min(DEPTH - 2, nsteps - 1 - st));
    __builtin_amdgcn_s_barrier();  // stage st visible to all waves; every wave is past compute(st-1)
    if (st + DEPTH - 1 < nsteps) {
      stage(fill);
      fill = (fill + 1 == DEPTH) ? 0 : fill + 1;
    }
    const char* base = ring + slot * STAGE_BYTES;
    if constexpr (FP8) {
      // The instruction's K order (tools/probes/mfma_scale_probe.hip, profiles/r05_mfma_scale_map.txt): a lane's first four
      // VGPRs are K elements 16 h .. 16 h + 15, its last four 64 + 16 h .. + 15, and the scale lane (l15, s) supplies governs
      // K elements 32 s .. 32 s + 31.  With 16-byte chunks h and 4 + h of the row (as in the bf16 form) the instruction's K
      // index IS the memory column, so MX block s of the K-step (columns 32 s ..) takes its scale from lane (l15, s).
      // (Chunks 2h, 2h + 1 - the r02 kernels' choice, equally good with unit scales - put half of every memory block under
      // another block's scale: first version of this kernel, caught by test_decode_proj_fp8_mx.)
      typedef int i32x8 __attribute__((ext_vector_type(8)));
      const int qlo = rd0, qhi = rd1;
      auto frag8 = [&](const char* p0) -> i32x8 {
        const u32x4 lo = *(const u32x4*)(p0 + qlo), hi = *(const u32x4*)(p0 + qhi);
        return (i32x8){(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
      };
      i32x8 xa[MB];
      int xs[MB];
      const char* sc = base + A_BYTES + DS_BN * DS_ROWB + wn * 256;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        xa[mb] = frag8(base + mb * 2048);
        // this lane's block (row 16 mb + l15, columns 32 h ..): byte h of the row's word, moved to byte 0
        xs[mb] = (int)((*(const uint32_t*)(sc + (mb * 16 + l15) * 4) >> (8 * h)) & 0xffu);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const i32x8 wa = frag8(base + w_rd + j * 2048);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
          acc[mb][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wa, xa[mb], acc[mb][j], 0, 0, 0, 0x7f, 0, xs[mb]);
      }
    } else {
      bf16x8 a0[MB], a1[MB];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        a0[mb] = *(const bf16x8*)(base + mb * 2048 + rd0);
        a1[mb] = *(const bf16x8*)(base + mb * 2048 + rd1);
      }
      bf16x8 wf[2][2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        wf[j][0] = *(const bf16x8*)(base + w_rd + j * 2048 + rd0);
        wf[j][1] = *(const bf16x8*)(base + w_rd + j * 2048 + rd1);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          acc[mb][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][0], a0[mb], acc[mb][j], 0, 0, 0);
          acc[mb][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][1], a1[mb], acc[mb][j], 0, 0, 0);
        }
    }
    slot = (slot + 1 == DEPTH) ? 0 : slot + 1;
    ++st;
  };

  // ------------------------------------------------------------------ the finished tile: lane holds
  // D[n = n0 + 32 wn + 16 j + 4 h + r][m = 16 mb + l15], r = 0..3
  auto epilogue = [&](f32x4 (&acc)[MB][2], int tile) __attribute__((always_inline)) {
    ds_epilogue<FP8, MB>(p, acc, 0, tile * DS_BN + wn * 32, rs, lane, red_s, wn);
  };

  // ------------------------------------------------------------------ one segment of this workgroup's range is complete
  const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc((void*)p.part, 0, 0x7fffffff, 0x00020000);
  constexpr unsigned BLOCK_BYTES = 16u * MB * DS_BN * 4u;   // one segment block: rows x 128 columns f32
  // byte offset of this lane's f32x4 (mb, j) inside a block: row-major [16 MB][128], a store instruction writes 4 rows' 16-byte
  // pieces... (rows differ by l15: 16 rows x 64 contiguous bytes per instruction)
  auto blk_off = [&](int mb, int j) -> unsigned { return (unsigned)(((mb * 16 + l15) * DS_BN + wn * 32 + 16 * j + 4 * h) * 4); };
  auto seg_geom = [&](int tile, int& first, int& ns) {
    first = (tile * p.nk_all) / p.spb;
    ns = ((tile + 1) * p.nk_all - 1) / p.spb - first + 1;
  };
  auto seg_block = [&](int tile, int first, int seg) -> unsigned {
    const int w = first + seg;
    const int start = max(tile * p.nk_all, w * p.spb);
    return (unsigned)ds_seg_id(start, p.nk_all, p.spb, p.lcm) * BLOCK_BYTES;
  };
  auto store_partial = [&](f32x4 (&acc)[MB][2], int tile) __attribute__((always_inline)) {
    int first, ns;
    seg_geom(tile, first, ns);
    if (ns == 1) return;
    const unsigned base = seg_block(tile, first, blockIdx.x - first);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int j = 0; j < 2; ++j) ds_store_sc1(prs, base + blk_off(mb, j), acc[mb][j]);
  };
  // sum of the tile's segments in segment order (this workgroup's own from registers), result in acc
  auto gather = [&](f32x4 (&acc)[MB][2], int tile, int first, int ns) __attribute__((always_inline)) {
    constexpr int GRP = G::GRP;
    const int mine = blockIdx.x - first;
    f32x4 sum[MB][2];
    for (int g0 = 0; g0 < ns; g0 += GRP) {
      f32x4 v[GRP][MB][2];
#pragma unroll
      for (int u = 0; u < GRP; ++u) {
        const int sg = min(g0 + u, ns - 1);          // clamped duplicate: loaded, never added
        const unsigned base = seg_block(tile, first, sg);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int j = 0; j < 2; ++j) v[u][mb][j] = ds_load_sc1(prs, base + blk_off(mb, j));
      }
#pragma unroll
      for (int u = 0; u < GRP; ++u) {
        const int sg = g0 + u;
        if (sg >= ns) break;                          // workgroup-uniform
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const f32x4 t = (sg == mine) ? acc[mb][j] : v[u][mb][j];
            if (sg == 0) sum[mb][j] = t;
            else { sum[mb][j][0] += t[0]; sum[mb][j][1] += t[1]; sum[mb][j][2] += t[2]; sum[mb][j][3] += t[3]; }
          }
      }
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[mb][j] = sum[mb][j];
  };

  // A workgroup's range touches a few tiles (gate/up: 65 steps over 56-step tiles = up to three).  Each gets its OWN
  // accumulator set and everything that leaves the registers does so after the last K-step: a store issued while the ring is
  // running sits in the same in-order-counted vmcnt queue as the ring's LDS-DMA loads (gemm_bf16.hip, r03 probe builds:
  // 20-23 % slower).  Ranges that touch more than NSEG tiles (small K: many short tiles; the lm_head's 4.6 tiles per
  // workgroup never do) finish the oldest set on the spot.
  f32x4 accs[NSEG][MB][2];
  int seg_tile[NSEG];

  // finish the sets [lo, hi): partial stores, ONE drain + barrier, tickets, then per set the epilogue where this workgroup
  // owns the tile or holds its last ticket.  The per-set part is ONE copy of the code working on a copy of the set (a
  // register array cannot be indexed by a run-time set number; 32 moves per set are nothing next to NSEG inlined epilogues)
  auto finish_sets = [&](int lo, int hi) __attribute__((always_inline)) {
    // the deferred RMSNorm's row factors: every unit partial of a row requested at once (lane = row, coalesced 256-byte rows);
    // the round trip hides under the partial stores' drain below.  (A first version computed rs in the kernel's prologue,
    // 8 tiles per round trip: 4 dependent round trips = ~8 us in front of the weight stream.)
    DsRowLoads sq;
    ds_row_factor_loads(p, sq, wn, lane);
#pragma unroll
    for (int sg = 0; sg < NSEG; ++sg)
      if (sg >= lo && sg < hi) {
        store_partial(accs[sg], seg_tile[sg]);
        if (tid == 0) segt_s[sg] = seg_tile[sg];
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave's stores are out (write-through) ...
    ds_row_factors(p, sq, red_s, rs_s, wn, lane);       // (two barriers: the second one is ...)
                                                        // ... before the lanes that signal for all of them
    if (tid >= lo && tid < hi) {                        // one lane per set: the tickets travel together (one round trip)
      const int tile = segt_s[tid];
      int first, ns;
      seg_geom(tile, first, ns);
      tick_s[tid] = (ns == 1) ? 0 : __hip_atomic_fetch_add(p.cnt + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) rs[mb] = rs_s[mb * 16 + l15];
    __syncthreads();                                    // the other waves load only behind this barrier
    // The per-set part exists once and always works on set 0; after every pass the sets move down by one.  (Selecting set
    // `sg` at run time - by index or by a chain of guarded copies, which hipcc folds back into an index - moves EVERY
    // accumulator set to scratch memory: 800 bytes per lane at MB = 4.  Static indices only.)
    if (lo != 0) {                                      // overflow path: the one live set is the last one
      seg_tile[0] = seg_tile[NSEG - 1];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) { accs[0][mb][0] = accs[NSEG - 1][mb][0]; accs[0][mb][1] = accs[NSEG - 1][mb][1]; }
    }
#pragma unroll 1
    for (int sg = lo; sg < hi; ++sg) {
      const int tile = seg_tile[0];
      int first, ns;
      seg_geom(tile, first, ns);
      bool mine = true;
      if (ns > 1) {
        mine = tick_s[sg] == ns - 1;                    // else somebody else will finish this tile (workgroup-uniform)
        if (mine) {
          gather(accs[0], tile, first, ns);
          if (tid == 0) __hip_atomic_store(p.cnt + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // next launch
        }
      }
      if (mine) epilogue(accs[0], tile);
#pragma unroll
      for (int k = 0; k + 1 < NSEG; ++k) {
        seg_tile[k] = seg_tile[k + 1];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) { accs[k][mb][0] = accs[k + 1][mb][0]; accs[k][mb][1] = accs[k + 1][mb][1]; }
      }
    }
    __syncthreads();                                    // tick_s is rewritten by a later call
  };

  int lo = 0, hi = 0;   // live sets
#pragma unroll
  for (int sg = 0; sg < NSEG; ++sg) {
    seg_tile[sg] = 0;
    if (st < nsteps) {   // workgroup-uniform
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        accs[sg][mb][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
        accs[sg][mb][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      hi = sg + 1;
      const int n_here = min(nsteps - st, p.nk_all - c_kt);
      for (int i = 0; i < n_here; ++i) step(accs[sg]);
      c_kt += n_here;
      seg_tile[sg] = c_tile;
      if (c_kt == p.nk_all) { c_kt = 0; ++c_tile; }
    }
  }
  // ONE call site of the finishing code (kept out of the unrolled loop above: with it inside, hipcc unrolls that loop only
  // partly and the accumulator sets land in scratch memory).  Ranges that touch more than NSEG tiles come round again: every
  // further tile goes through the last set and is finished on the spot (drains the ring once per tile).
  for (;;) {
    finish_sets(lo, hi);
    if (st >= nsteps) break;
    lo = NSEG - 1;
    hi = NSEG;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      accs[NSEG - 1][mb][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
      accs[NSEG - 1][mb][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const int n_here = min(nsteps - st, p.nk_all - c_kt);
    for (int i = 0; i < n_here; ++i) step(accs[NSEG - 1]);
    c_kt += n_here;
    seg_tile[NSEG - 1] = c_tile;
    if (c_kt == p.nk_all) { c_kt = 0; ++c_tile; }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// geometry of the stream-K cut for (N, K): steps per workgroup, workgroups, and a bound on the segments of split tiles
static long ds_gcd(long a, long b) { while (b) { const long t = a % b; a = b; b = t; } return a; }
static void ds_geometry(int N, int nk, int* spb, int* nwg, int* lcm, int* nblocks) {
  const int tiles = (N + DS_BN - 1) / DS_BN;
  const long total = (long)tiles * nk;
  int wg = DS_MAX_WG;
  for (;;) {
    long per = (total + wg - 1) / wg;
    if (per < 4) per = total < 4 ? total : 4;  // never cut finer than 4 K-steps
    int slots = 1;
    for (int t = 0; t < tiles; ++t) {
      const int s = (int)(((long)(t + 1) * nk - 1) / per - ((long)t * nk) / per) + 1;
      if (s > slots) slots = s;
    }
    if (slots <= DS_MAX_SEGS || wg == 1) {
      *spb = (int)per;
      *nwg = (int)((total + per - 1) / per);
      long l = (long)nk / ds_gcd(nk, per) * per;
      if (l > total) l = total + 1;              // no common multiple inside the sequence
      *lcm = (int)l;
      *nblocks = tiles + *nwg;                   // every seam starts one segment
      return;
    }
    wg = wg / 2;  // fewer, longer ranges -> fewer segments per tile
  }
}

static int ds_rows(int B) { return B <= 16 ? 16 : B <= 32 ? 32 : 64; }

// workspace of vis_decode_proj_*: [DS_CNT_BYTES of arrival counters][segment blocks]; zero it once (the kernel leaves the
// counters at zero; the blocks need no initialisation); one launch at a time per workspace; a workspace sized for the
// largest (B, N, K) of a model serves all its projections
extern "C" long long vis_decode_proj_ws_bytes(int B, int N, int K, int fp8) {
  if (B <= 0 || B > 64 || N <= 0 || K <= 0) return 0;
  const int kstep = fp8 ? 128 : 64;
  if (K % kstep) return 0;
  int spb, nwg, lcm, nblocks;
  ds_geometry(N, K / kstep, &spb, &nwg, &lcm, &nblocks);
  if ((N + DS_BN - 1) / DS_BN > DS_CNT_BYTES / 4) return 0;
  return DS_CNT_BYTES + (long long)nblocks * ds_rows(B) * DS_BN * 4;
}

template <bool FP8>
static int ds_launch(DsArgs p, int B, int K, void* ws, hipStream_t stream) {
  const int kstep = FP8 ? 128 : 64;
  int spb, nwg, lcm, nblocks;
  p.nk_all = K / kstep;
  ds_geometry(p.N, p.nk_all, &spb, &nwg, &lcm, &nblocks);
  const int tiles = (p.N + DS_BN - 1) / DS_BN;
  p.total = tiles * p.nk_all;
  p.spb = spb;
  p.lcm = lcm;
  if (tiles > DS_CNT_BYTES / 4) return VIS_ERR_ARG;
  p.cnt = (int*)ws;
  p.part = (float*)((char*)ws + DS_CNT_BYTES);
  static const bool attr_ok = [] {
    return hipFuncSetAttribute((const void*)decode_proj_kernel<FP8, 1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               DsGeom<FP8, 1>::LDS_BYTES) == hipSuccess &&
           hipFuncSetAttribute((const void*)decode_proj_kernel<FP8, 2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               DsGeom<FP8, 2>::LDS_BYTES) == hipSuccess &&
           hipFuncSetAttribute((const void*)decode_proj_kernel<FP8, 4>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               DsGeom<FP8, 4>::LDS_BYTES) == hipSuccess;
  }();
  if (!attr_ok) return VIS_ERR_LAUNCH;
  vis_clear_error();
  if (B > 32) hipLaunchKernelGGL((decode_proj_kernel<FP8, 4>), dim3(nwg), dim3(256), (DsGeom<FP8, 4>::LDS_BYTES), stream, p);
  else if (B > 16) hipLaunchKernelGGL((decode_proj_kernel<FP8, 2>), dim3(nwg), dim3(256), (DsGeom<FP8, 2>::LDS_BYTES), stream, p);
  else hipLaunchKernelGGL((decode_proj_kernel<FP8, 1>), dim3(nwg), dim3(256), (DsGeom<FP8, 1>::LDS_BYTES), stream, p);
  return vis_check_launch();
}

// Batched decode projection with its epilogue in the same launch (see the top of this file).
//   mode VIS_DP_PLAIN       C[b][n]   = (x W^T)[b][n] * rs[b] + bias[n]                      bf16, or f32 when out_f32
//   mode VIS_DP_SWIGLU      C[b][o]   = silu(g * rs[b]) * (u * rs[b]) over the 16-row interleaved gate/up weight (N / 2 outputs)
//   mode VIS_DP_RESID_NORMW C = y = bf16(x W^T + R);  Cw = bf16(y * nw[n]);  ssq_out[n / 32][b] = sum of y^2 over the 32-column unit
// rs[b] = rsqrt(sum_u ssq_in[u][b] / norm_dim + eps) when ssq_in != NULL (tiles_in = norm_dim / 32 units <= 128), else 1.
// Cq / Cqs (SWIGLU, RESID_NORMW; may be NULL): the row the NEXT projection consumes (act, or y * nw) as MX blocks - e4m3
// bytes [B][ldcq] + one E8M0 scale byte per 32 columns [B][ldcqs] - for a following vis_decode_proj_fp8.
extern "C" int vis_decode_proj_bf16(const void* A, const void* W, void* ws, void* C, void* Cw, void* Cq, void* Cqs,
                                    const void* bias, const void* R, const void* nw, const void* ssq_in, void* ssq_out,
                                    int B, int N, int K, int lda, int ldw, int ldc, int ldr, int ldcq, int ldcqs, int mode,
                                    int out_f32, int tiles_in, int norm_dim, float eps, hipStream_t stream) {
  DsArgs p = {};
  p.A = (const char*)A; p.W = (const char*)W; p.C = C; p.Cw = (bf16_t*)Cw; p.Cq = (uint8_t*)Cq; p.Cqs = (uint8_t*)Cqs;
  p.bias = (const bf16_t*)bias; p.R = (const bf16_t*)R; p.nw = (const bf16_t*)nw; p.ssq_in = (const float*)ssq_in;
  p.ssq_out = (float*)ssq_out;
  p.M = B; p.N = N; p.lda = lda * 2; p.ldw = ldw * 2; p.ldc = ldc; p.ldr = ldr; p.ldcq = ldcq; p.ldcqs = ldcqs;
  p.mode = mode; p.out_f32 = out_f32; p.tiles_in = tiles_in;
  p.inv_norm_dim = norm_dim > 0 ? 1.0f / (float)norm_dim : 0.f; p.eps = eps;
  const int rc = ds_check_common(p, B, K, ws, mode, false);
  if (rc != VIS_OK) return rc;
  if (lda % 8 != 0 || ldw % 8 != 0 || lda < K || ldw < K || (ssq_in && norm_dim <= 0)) return VIS_ERR_ARG;
  return ds_launch<false>(p, B, K, ws, stream);
}

// fp8 form (BASELINE configs[4]): A = MX blocks (Aq e4m3 bytes [B][ldaq] + As E8M0 scales [B][ldas], one per 32 columns),
// Wq e4m3 [N][ldw] with per-row f32 scales sw[N]; epilogue as above with (x W^T)[b][n] * sw[n] in place of the raw sum.
// Cq / Cqs (SWIGLU, RESID_NORMW): the row the NEXT projection consumes (act, or y * nw) as MX blocks; K % 128 == 0.
extern "C" int vis_decode_proj_fp8(const void* Aq, const void* As, const void* Wq, const void* sw, void* ws, void* C,
                                   void* Cw, void* Cq, void* Cqs, const void* bias, const void* R, const void* nw,
                                   const void* ssq_in, void* ssq_out, int B, int N, int K, int ldaq, int ldas, int ldw,
                                   int ldc, int ldr, int ldcq, int ldcqs, int mode, int out_f32, int tiles_in,
                                   int norm_dim, float eps, hipStream_t stream) {
  DsArgs p = {};
  p.A = (const char*)Aq; p.As = (const uint8_t*)As; p.W = (const char*)Wq; p.sw = (const float*)sw;
  p.C = C; p.Cw = (bf16_t*)Cw; p.Cq = (uint8_t*)Cq; p.Cqs = (uint8_t*)Cqs; p.bias = (const bf16_t*)bias;
  p.R = (const bf16_t*)R; p.nw = (const bf16_t*)nw; p.ssq_in = (const float*)ssq_in; p.ssq_out = (float*)ssq_out;
  p.M = B; p.N = N; p.lda = ldaq; p.ldas = ldas; p.ldw = ldw; p.ldc = ldc; p.ldr = ldr; p.ldcq = ldcq; p.ldcqs = ldcqs;
  p.mode = mode; p.out_f32 = out_f32; p.tiles_in = tiles_in;
  p.inv_norm_dim = norm_dim > 0 ? 1.0f / (float)norm_dim : 0.f; p.eps = eps;
  const int rc = ds_check_common(p, B, K, ws, mode, true);
  if (rc != VIS_OK) return rc;
  if (!As || !sw || ldaq % 16 != 0 || ldw % 16 != 0 || ldaq < K || ldw < K || ldas % 4 != 0 || ldas < K / 32 ||
      ((uintptr_t)As & 3) || ((uintptr_t)sw & 15) || (ssq_in && norm_dim <= 0))
    return VIS_ERR_ARG;
  return ds_launch<true>(p, B, K, ws, stream);
}

// ---------------------------------------------------------------------------------------------------------------------
// Head of a batched decode step: x[b] = table[ids[b]] (embedding lookup, clamped like vis_gather_rows); xw = bf16(x * nw);
// ssq[n / 32][b] = the 32-column unit's sum of x^2 - the operands the first projection (vis_decode_proj_*) expects; optionally
// the MX copy of xw.  One workgroup per sequence.
struct PrepArgs {
  const bf16_t* table;
  const int* ids;
  const bf16_t* nw;
  bf16_t* x;
  bf16_t* xw;
  uint8_t* xq;
  uint8_t* xqs;
  float* ssq;
  int rows, H, ldx, ldq, ldqs;
};

__global__ __launch_bounds__(256) void decode_prep_rows_kernel(PrepArgs p) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const int id = min(max(p.ids[b], 0), p.rows - 1);
  const bf16_t* src = p.table + (size_t)id * p.H;
  // 8 columns per thread and pass; a 32-column unit (= an MX block) = 4 consecutive threads
  for (int c0 = 0; c0 < p.H; c0 += 256 * 8) {
    const int c = c0 + tid * 8;
    const bool ok = c < p.H;
    float f[8], g[8], o[8];
    const u32x4 z = (u32x4){0u, 0u, 0u, 0u};
    const u32x4 raw = ok ? *(const u32x4*)(src + c) : z;
    unpack8(raw, f);
    unpack8(ok ? *(const u32x4*)(p.nw + c) : z, g);
    float ss = 0.f, am = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      ss += f[e] * f[e];
      o[e] = bf2f(f2bf(f[e] * g[e]));
      am = fmaxf(am, fabsf(o[e]));
    }
    if (ok) {
      *(u32x4*)(p.x + (size_t)b * p.ldx + c) = raw;
      if (p.xw) *(u32x4*)(p.xw + (size_t)b * p.ldx + c) = pack8(o);
    }
    // unit sums (32 columns = 4 lanes of one quad)
    ss += VIS_DPP(ss, 0xB1);
    ss += VIS_DPP(ss, 0x4E);
    if (ok && (tid & 3) == 0) p.ssq[(size_t)(c >> 5) * DS_SSQ_LD + b] = ss;
    if (p.xq) {
      am = fmaxf(am, VIS_DPP(am, 0xB1));
      am = fmaxf(am, VIS_DPP(am, 0x4E));
      const int sb = ds_mx_scale_byte(am);
      const float inv = ds_mx_inv_scale(sb);
      if (ok) {
        int w0 = 0, w1 = 0;
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(o[0] * inv, o[1] * inv, w0, false);
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(o[2] * inv, o[3] * inv, w0, true);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(o[4] * inv, o[5] * inv, w1, false);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(o[6] * inv, o[7] * inv, w1, true);
        *(u32x2*)(p.xq + (size_t)b * p.ldq + c) = (u32x2){(uint32_t)w0, (uint32_t)w1};
        if ((tid & 3) == 0) p.xqs[(size_t)b * p.ldqs + (c >> 5)] = (uint8_t)sb;
      }
    }
  }
}

extern "C" int vis_decode_prep_rows(const void* table, const void* ids, const void* nw, void* x, void* xw, void* xq,
                                    void* xqs, void* ssq, int B, int table_rows, int H, int ldx, int ldq, int ldqs,
                                    hipStream_t stream) {
  if (!table || !ids || !nw || !x || !ssq || B <= 0 || B > 64 || table_rows <= 0 || H <= 0) return VIS_ERR_ARG;
  if (H % 128 != 0 || ldx % 8 != 0 || ldx < H || (!xw && !xq) || (xq != nullptr) != (xqs != nullptr)) return VIS_ERR_ARG;
  if (xq && (ldq % 8 != 0 || ldq < H || ldqs < H / 32)) return VIS_ERR_ARG;
  if (((uintptr_t)table | (uintptr_t)nw | (uintptr_t)x | (uintptr_t)xw) & 15 || ((uintptr_t)xq & 7) || ((uintptr_t)ssq & 3))
    return VIS_ERR_ARG;
  PrepArgs p;
  p.table = (const bf16_t*)table; p.ids = (const int*)ids; p.nw = (const bf16_t*)nw; p.x = (bf16_t*)x; p.xw = (bf16_t*)xw;
  p.xq = (uint8_t*)xq; p.xqs = (uint8_t*)xqs; p.ssq = (float*)ssq;
  p.rows = table_rows; p.H = H; p.ldx = ldx; p.ldq = ldq; p.ldqs = ldqs;
  vis_clear_error();
  hipLaunchKernelGGL(decode_prep_rows_kernel, dim3(B), dim3(256), 0, stream, p);
  return vis_check_launch();
}
