// K10 (batched decode), column-parallel form: every workgroup owns a slab of whole output columns and the ENTIRE K, so no
// partial sums ever cross a workgroup - no slabs, no tickets, no finalisation launch, and the epilogue (bias / SwiGLU /
// residual + next norm weight + sums of squares / MX blocks, decode_proj_common.hip.h) runs on accumulators that never left
// the registers.
//
// Why a second form next to decode_stream.hip.  The stream-K form balances the weight bytes perfectly but has to reduce cut
// tiles across workgroups, and a cross-CU reduction is three dependent memory round trips (store + drain, ticket, gather)
// however it is arranged: measured at 64 sequences (profiles/r05_decode_step_b64_streamk.txt) the in-kernel last-arriver
// reduction cost MORE than the r04 pair of launches it replaced - qkv 21 vs 11 + 5 us, o 24 vs 10 + 5, down 64 vs 33 + 5 -
// because only the 28-36 workgroups that hold a tile's last ticket do the gathering (224 KB each at ~60 GB/s per block,
// MI355X_MICROARCH.md "handoff-payload").  What the reduction buys is K-parallelism for projections with few column tiles;
// what it costs is fixed.  Here the parallelism comes from the COLUMNS instead: N / 32 units of 32 columns (one MX block,
// one SwiGLU gate/up group, two MFMA column blocks) are dealt to min(256, N / 32) workgroups, floor or ceil(units / workgroups)
// each, at most five.  The price is the activation tile: every workgroup streams all of x (64 rows x K) from L2 - 458 KB at
// K = 3584, bf16, 64 sequences - next to its own 32 x K weight rows from HBM, which is why the long-K down projection
// (x = 2.4 MB per workgroup at 64 sequences) stays on the split-K forms and everything else comes here.
//
//  * ring: one stage = x tile (32 or 64 rows x 128 B) + `cnt` weight units (32 rows x 128 B each) [+ the K-step's MX scale
//    words, fp8]; the stage size and with it the ring depth are run-time (a one-unit workgroup keeps 10 stages of 12 KB in
//    flight, a five-unit one 5 stages of 28 KB); counted vmcnt through a jump table (the count is a run-time product).
//  * a wave owns one unit (wave 0 a second one in five-unit workgroups: the MFMA is not the bound); at 33-64 sequences a
//    one-unit workgroup splits the ROWS over its waves instead (wave w = rows 16 w .. 16 w + 15): 16 MFMAs per K-step on one
//    wave would be as long as the step's DMA.
//  * K order per output element: ascending K-steps, the two 32-deep halves of a step in order - the same as the stream-K
//    form's sole-owner tiles; a row's result depends on that row alone.
#include "decode_proj_common.hip.h"

#define CP_MAX_UNITS 5
#define CP_MAX_DEPTH 10
#define CP_RING_BYTES (150 * 1024)

template <bool FP8, int MB>
struct CpGeom {
  static constexpr int XI = (MB == 4) ? 2 : 1;        // x LDS-DMA instructions per thread per stage
  static constexpr int A_BYTES = XI * 32 * DS_ROWB;   // x tile: 32 (MB = 4: 64) rows
  static constexpr int S_BYTES = FP8 ? 4 * 64 * 4 : 0;
};

__device__ __forceinline__ void cp_wait_vmcnt(int n) {   // s_waitcnt vmcnt(n), n a run-time value in 0..63 (workgroup-uniform)
#define CP_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
  switch (n) {
    CP_W(0) CP_W(1) CP_W(2) CP_W(3) CP_W(4) CP_W(5) CP_W(6) CP_W(7) CP_W(8) CP_W(9) CP_W(10) CP_W(11) CP_W(12) CP_W(13) CP_W(14)
    CP_W(15) CP_W(16) CP_W(17) CP_W(18) CP_W(19) CP_W(20) CP_W(21) CP_W(22) CP_W(23) CP_W(24) CP_W(25) CP_W(26) CP_W(27) CP_W(28)
    CP_W(29) CP_W(30) CP_W(31) CP_W(32) CP_W(33) CP_W(34) CP_W(35) CP_W(36) CP_W(37) CP_W(38) CP_W(39) CP_W(40) CP_W(41) CP_W(42)
    CP_W(43) CP_W(44) CP_W(45) CP_W(46) CP_W(47) CP_W(48) CP_W(49) CP_W(50) CP_W(51) CP_W(52) CP_W(53) CP_W(54) CP_W(55) CP_W(56)
    CP_W(57) CP_W(58) CP_W(59) CP_W(60) CP_W(61) CP_W(62)
    default: asm volatile("s_waitcnt vmcnt(63)" ::: "memory"); break;
  }
#undef CP_W
}

template <bool FP8, int MB>
__global__ __launch_bounds__(256, 1) void decode_colpar_kernel(DsArgs p, int units, int nk) {
  using G = CpGeom<FP8, MB>;
  constexpr int XI = G::XI, A_BYTES = G::A_BYTES;
  extern __shared__ __attribute__((aligned(16))) char ring[];
  __shared__ float rs_s[64];
  __shared__ float part_s[4][64];
  const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
  const int l15 = lane & 15, h = lane >> 4;
  // this workgroup's units: floor or ceil(units / workgroups) consecutive ones
  const int nwg = gridDim.x, base_cnt = units / nwg, extra = units - base_cnt * nwg;
  const int w = blockIdx.x;
  const int u0 = w * base_cnt + min(w, extra), cnt = base_cnt + (w < extra ? 1 : 0);   // 1 .. CP_MAX_UNITS
  const int col_base = u0 * 32;
  const bool narrow = (MB == 4) && cnt == 1;           // rows over the waves instead of units over the waves

  // the deferred RMSNorm's row partials: requested first, consumed after the stream
  DsRowLoads sq;
  ds_row_factor_loads(p, sq, wn, lane);

  const int w_bytes = cnt * 4096;
  const int stage_bytes = A_BYTES + w_bytes + G::S_BYTES;
  const int per = XI + cnt + (FP8 ? 1 : 0);            // vm instructions per thread per stage
  int depth = CP_RING_BYTES / stage_bytes;
  if (depth > CP_MAX_DEPTH) depth = CP_MAX_DEPTH;
  if ((depth - 2) * per > 63) depth = 63 / per + 2;    // vmcnt is a 6-bit counter

  // ---- staging offsets: x tile rows (rows >= M repeat row M - 1), one weight unit per instruction (32 rows x 8 chunks)
  uint32_t a_off[XI], w_off[CP_MAX_UNITS], s_off = 0;
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int row = i * 32 + (tid >> 3), ch = (tid & 7) ^ (row & 7);
    a_off[i] = (uint32_t)min(row, p.M - 1) * (uint32_t)p.lda + ch * 16;
  }
  {
    const int row = tid >> 3, ch = (tid & 7) ^ (row & 7);
#pragma unroll
    for (int k = 0; k < CP_MAX_UNITS; ++k)
      w_off[k] = (uint32_t)(min(col_base + 32 * min(k, cnt - 1) + row, p.N - 1)) * (uint32_t)p.ldw + ch * 16;
  }
  if (FP8) s_off = (uint32_t)min(lane, p.M - 1) * (uint32_t)p.ldas;
  const char* a_ptr = p.A;
  const char* w_ptr = p.W;
  const char* s_ptr = FP8 ? (const char*)p.As : nullptr;
  const int wave_base = __builtin_amdgcn_readfirstlane(tid >> 6) * 1024;

  auto stage = [&](int slot) __attribute__((always_inline)) {
    char* base = ring + slot * stage_bytes + wave_base;
#pragma unroll
    for (int i = 0; i < XI; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_ptr + a_off[i]),
                                       (__attribute__((address_space(3))) void*)(base + i * 4096), 16, 0, 0);
#pragma unroll
    for (int k = 0; k < CP_MAX_UNITS; ++k)
      if (k < cnt)   // workgroup-uniform
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_ptr + w_off[k]),
                                         (__attribute__((address_space(3))) void*)(base + A_BYTES + k * 4096), 16, 0, 0);
    if (FP8)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s_ptr + s_off),
                                       (__attribute__((address_space(3))) void*)(ring + slot * stage_bytes + A_BYTES + w_bytes + (wave_base >> 2)),
                                       4, 0, 0);
    a_ptr += DS_ROWB;
    w_ptr += DS_ROWB;
    if (FP8) s_ptr += 4;
  };

  const int sw7 = lane & 7;
  const int rd0 = l15 * 128 + (((0 + h) ^ sw7) << 4);
  const int rd1 = l15 * 128 + (((4 + h) ^ sw7) << 4);

  // accumulators: unit slot 0 (this wave's unit, or in the narrow form this wave's row block) and slot 1 (wave 0's second unit)
  f32x4 acc0[MB][2], acc1[MB][2];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int j = 0; j < 2; ++j) { acc0[mb][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc1[mb][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  const bool has0 = narrow || wn < cnt;
  const bool has1 = !narrow && wn == 0 && cnt == CP_MAX_UNITS;

  // one K-step of unit `unit` into acc, row blocks [mb_lo, mb_lo + NMB)
  auto mfma_unit = [&](const char* sbase, int unit, f32x4 (&acc)[MB][2], int mb_lo, int nmb, int wave_slot) __attribute__((always_inline)) {
    const char* wb = sbase + A_BYTES + unit * 4096;
    if constexpr (FP8) {
      typedef int i32x8 __attribute__((ext_vector_type(8)));
      auto frag8 = [&](const char* p0) -> i32x8 {   // chunks h and 4 + h: the instruction's K index is the memory column
        const u32x4 lo = *(const u32x4*)(p0 + rd0), hi = *(const u32x4*)(p0 + rd1);
        return (i32x8){(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
      };
      const char* sc = sbase + A_BYTES + w_bytes + wave_slot * 256;
      const i32x8 wa0 = frag8(wb), wa1 = frag8(wb + 2048);
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        if (mb >= nmb) break;
        const int row_blk = mb_lo + mb;
        const i32x8 xa = frag8(sbase + row_blk * 2048);
        const int xs = (int)((*(const uint32_t*)(sc + (row_blk * 16 + l15) * 4) >> (8 * h)) & 0xffu);
        acc[mb][0] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wa0, xa, acc[mb][0], 0, 0, 0, 0x7f, 0, xs);
        acc[mb][1] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wa1, xa, acc[mb][1], 0, 0, 0, 0x7f, 0, xs);
      }
    } else {
      bf16x8 wf[2][2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        wf[j][0] = *(const bf16x8*)(wb + j * 2048 + rd0);
        wf[j][1] = *(const bf16x8*)(wb + j * 2048 + rd1);
      }
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        if (mb >= nmb) break;
        const int row_blk = mb_lo + mb;
        const bf16x8 a0 = *(const bf16x8*)(sbase + row_blk * 2048 + rd0), a1 = *(const bf16x8*)(sbase + row_blk * 2048 + rd1);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[mb][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][0], a0, acc[mb][j], 0, 0, 0);
          acc[mb][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][1], a1, acc[mb][j], 0, 0, 0);
        }
      }
    }
  };

  // ---- the stream
  const int nsteps = nk;
  const int pre = min(depth - 1, nsteps);
  for (int s = 0; s < pre; ++s) stage(s);
  int slot = 0, fill = pre % depth;
  for (int st = 0; st < nsteps; ++st) {
    cp_wait_vmcnt(min(depth - 2, nsteps - 1 - st) * per);
    __builtin_amdgcn_s_barrier();   // stage st visible to all waves; every wave is past compute(st - 1)
    if (st + depth - 1 < nsteps) {
      stage(fill);
      fill = (fill + 1 == depth) ? 0 : fill + 1;
    }
    const char* sbase = ring + slot * stage_bytes;
    if (narrow) {
      mfma_unit(sbase, 0, acc0, wn, 1, wn);          // wave w: row block w of the one unit (its own copy of the scale words)
    } else {
      if (has0) mfma_unit(sbase, wn, acc0, 0, MB, wn);
      if (has1) mfma_unit(sbase, CP_MAX_UNITS - 1, acc1, 0, MB, wn);
    }
    slot = (slot + 1 == depth) ? 0 : slot + 1;
  }

  // ---- epilogue on the registers
  ds_row_factors(p, sq, part_s, rs_s, wn, lane);
  if (narrow) {
    if constexpr (MB == 4) {
      f32x4 a1[1][2] = {{acc0[0][0], acc0[0][1]}};
      const float r1[1] = {rs_s[wn * 16 + l15]};
      ds_epilogue<FP8, 1>(p, a1, wn, col_base, r1, lane, nullptr, 0);
    }
    return;
  }
  float rs[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) rs[mb] = rs_s[mb * 16 + l15];
  if (has0) ds_epilogue<FP8, MB>(p, acc0, 0, col_base + wn * 32, rs, lane, nullptr, 0);
  if (has1) ds_epilogue<FP8, MB>(p, acc1, 0, col_base + (CP_MAX_UNITS - 1) * 32, rs, lane, nullptr, 0);
}

// units per workgroup for (N): 0 = the column-parallel form does not cover the shape
static int cp_geometry(int N, int* units, int* nwg) {
  if (N <= 0 || N % 32 != 0) return 0;
  *units = N / 32;
  *nwg = *units < 256 ? *units : 256;
  const int per_wg = (*units + *nwg - 1) / *nwg;
  return per_wg <= CP_MAX_UNITS ? per_wg : 0;
}

template <bool FP8>
static int cp_launch(DsArgs p, int B, int K, hipStream_t stream) {
  int units, nwg;
  if (!cp_geometry(p.N, &units, &nwg)) return VIS_ERR_UNSUPPORTED;
  if (p.mode == DS_SWIGLU && p.Cq) return VIS_ERR_UNSUPPORTED;   // an MX block of act columns spans two units: stream-K form
  const int nk = K / (FP8 ? 128 : 64);
  const int lds = CP_RING_BYTES + 4096;     // run-time stage size: the whole ring budget (+ slack for the largest stage's tail)
  static const bool attr_ok = [] {
    const int l = CP_RING_BYTES + 4096;
    return hipFuncSetAttribute((const void*)decode_colpar_kernel<FP8, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, l) == hipSuccess &&
           hipFuncSetAttribute((const void*)decode_colpar_kernel<FP8, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, l) == hipSuccess &&
           hipFuncSetAttribute((const void*)decode_colpar_kernel<FP8, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, l) == hipSuccess;
  }();
  if (!attr_ok) return VIS_ERR_LAUNCH;
  vis_clear_error();
  if (B > 32) hipLaunchKernelGGL((decode_colpar_kernel<FP8, 4>), dim3(nwg), dim3(256), lds, stream, p, units, nk);
  else if (B > 16) hipLaunchKernelGGL((decode_colpar_kernel<FP8, 2>), dim3(nwg), dim3(256), lds, stream, p, units, nk);
  else hipLaunchKernelGGL((decode_colpar_kernel<FP8, 1>), dim3(nwg), dim3(256), lds, stream, p, units, nk);
  return vis_check_launch();
}

// 1: vis_decode_proj_colpar_* covers (N, mode, MX output) - N a multiple of 32, at most five 32-column units per workgroup
// (N <= 40960), not SwiGLU with an MX output; 0: use vis_decode_proj_* (stream-K form).
extern "C" int vis_decode_proj_colpar_covers(int N, int mode, int mx_out) {
  int units, nwg;
  if (!cp_geometry(N, &units, &nwg)) return 0;
  return !(mode == DS_SWIGLU && mx_out);
}

// Column-parallel form of vis_decode_proj_bf16 (same arguments minus the workspace, same epilogues, same per-row results up to
// the summation order of the K-steps' partial products - which is the stream-K form's for tiles it does not cut).
// VIS_ERR_UNSUPPORTED for shapes it does not cover (vis_decode_proj_colpar_covers).
extern "C" int vis_decode_proj_colpar_bf16(const void* A, const void* W, void* C, void* Cw, void* Cq, void* Cqs, const void* bias,
                                           const void* R, const void* nw, const void* ssq_in, void* ssq_out, int B, int N, int K,
                                           int lda, int ldw, int ldc, int ldr, int ldcq, int ldcqs, int mode, int out_f32,
                                           int tiles_in, int norm_dim, float eps, hipStream_t stream) {
  DsArgs p = {};
  p.A = (const char*)A; p.W = (const char*)W; p.C = C; p.Cw = (bf16_t*)Cw; p.Cq = (uint8_t*)Cq; p.Cqs = (uint8_t*)Cqs;
  p.bias = (const bf16_t*)bias; p.R = (const bf16_t*)R; p.nw = (const bf16_t*)nw; p.ssq_in = (const float*)ssq_in;
  p.ssq_out = (float*)ssq_out;
  p.M = B; p.N = N; p.lda = lda * 2; p.ldw = ldw * 2; p.ldc = ldc; p.ldr = ldr; p.ldcq = ldcq; p.ldcqs = ldcqs;
  p.mode = mode; p.out_f32 = out_f32; p.tiles_in = tiles_in;
  p.inv_norm_dim = norm_dim > 0 ? 1.0f / (float)norm_dim : 0.f; p.eps = eps;
  const int rc = ds_check_common(p, B, K, A /* no workspace: any aligned non-null pointer */, mode, false);
  if (rc != VIS_OK) return rc;
  if (lda % 8 != 0 || ldw % 8 != 0 || lda < K || ldw < K || (ssq_in && norm_dim <= 0)) return VIS_ERR_ARG;
  return cp_launch<false>(p, B, K, stream);
}

extern "C" int vis_decode_proj_colpar_fp8(const void* Aq, const void* As, const void* Wq, const void* sw, void* C, void* Cw,
                                          void* Cq, void* Cqs, const void* bias, const void* R, const void* nw,
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
  const int rc = ds_check_common(p, B, K, Aq, mode, true);
  if (rc != VIS_OK) return rc;
  if (!As || !sw || ldaq % 16 != 0 || ldw % 16 != 0 || ldaq < K || ldw < K || ldas % 4 != 0 || ldas < K / 32 ||
      ((uintptr_t)As & 3) || ((uintptr_t)sw & 15) || (ssq_in && norm_dim <= 0))
    return VIS_ERR_ARG;
  return cp_launch<true>(p, B, K, stream);
}
