// K4: rotary embedding + head split (+ KV-cache write, + V transpose) for gfx950.
//
// One kernel serves both towers:
//  * LLM M-RoPE  (TF:models/qwen2_vl/modeling_qwen2_vl.py:180-222): the host
//    builds the per-token cos/sin rows [S, head_dim] f32 with the (t,h,w)
//    section already selected per channel (exactly what
//    apply_multimodal_rotary_pos_emb assembles from cos.split(mrope_section*2));
//  * ViT 2-D RoPE (TF:...:225-236): cos/sin rows [N, 80] f32 from (h,w) ids.
// In both cases  out = x*cos + rotate_half(x)*sin  over the full head_dim,
// computed in f32 and rounded to bf16 once.
//
// Input  : packed projection rows  qkv[S, (Hq+2*Hkv)*HD]  (bias already added
//          by the GEMM epilogue).
// Outputs: Q  [Hq ][S][HD]                     (rotated, head-major for attention)
//          (Hq == 0 or Hkv == 0 select a k/v-only or q-only split; cos == NULL skips the rotation)
//          K  [Hkv][k_tokens][HD] at row k_pos0+s (rotated; the LLM passes its
//             KV-cache layer here so prefill writes the cache in place)
//          V  [Hkv][k_tokens][HD] at row k_pos0+s (optional row-major copy: KV cache)
//          Vt [Hkv][HD][vt_ld]   (optional; keys contiguous - the PV operand
//             layout of the prefill attention kernel; pad columns zeroed).
//             Column order inside every aligned group of 32 keys: key 16 a + 4 h + r (a = 0..1, h = 0..3,
//             r = 0..3) sits at column 8 h + 4 a + r - the k-slot order the attention kernel's P fragment
//             has (two S^T accumulator blocks packed in-lane), so that a lane's 8 V values are ONE aligned
//             16-byte chunk (a single ds_read_b128) instead of two 8-byte pieces 32 bytes apart.
// HBM-bound: every qkv element is read once and written once (V twice).
#include "common.hip.h"

struct RopeArgs {
  const bf16_t* qkv;
  const float* cosv;
  const float* sinv;
  bf16_t* q;
  bf16_t* k;
  bf16_t* v;
  bf16_t* vt;
  int S, ld_qkv, Hq, Hkv;
  int k_tokens;  // rows per head in k/v outputs
  int k_pos0;    // first row written in k/v outputs
  int vt_ld;     // row stride of Vt (multiple of 64, >= round_up(S, 64))
  ReqOffsets req; // grid.z > 1: request blockIdx.z reads qkv + z * qkv_bs and writes q + z * q_bs, k / v + kv[z], vt + z * vt_bs
};

// Work decomposition (r02): the cos / sin rows are f32 and four times the bytes of the bf16 data they rotate, and the
// first version (one workgroup per (64 tokens, head)) re-read them for every head: 2.9 TB/s of useful traffic.  Now
//  * "rope" workgroups own 16 tokens and ALL q / k heads: a thread holds the cos / sin values of one (token, 8-dim
//    pair-chunk) in registers and walks the heads (in HG interleaved groups), so a table row is read once per token;
//  * "V" workgroups own (64 tokens, one kv head): row-major copy into the cache and the transposed tile through LDS
//    (64 tokens = one 128-byte run of V^T columns).
// Both kinds live in ONE launch: blocks [0, n_rope) are rope blocks, the rest V blocks.
template <int HD>
__global__ __launch_bounds__(256) void qkv_rope_split_kernel(RopeArgs p, int n_rope) {
  constexpr int HALF = HD / 2;
  constexpr int PC = HD / 16;  // pair-chunks per (token, head): 8 dims + their rotate_half partners
  constexpr int CH = HD / 8;   // 16-byte chunks per (token, head)
  constexpr int VT_LD = 64 + 8;
  constexpr int HG = 256 / (16 * PC);   // head groups per workgroup: 2 (head_dim 128) / 3 (head_dim 80)
  __shared__ __attribute__((aligned(16))) bf16_t tile[HD * VT_LD];
  const int tid = threadIdx.x;
  // (locals, not updates of `p`: writing to a by-value kernel argument makes the compiler keep the whole struct in scratch)
  const int z = blockIdx.z;
  const long long kvo = req_kv(p.req, z);
  const bf16_t* const r_qkv = p.qkv + z * p.req.qkv_bs;
  bf16_t* const r_q = p.q ? p.q + z * p.req.q_bs : nullptr;
  bf16_t* const r_k = p.k ? p.k + kvo : nullptr;
  bf16_t* const r_v = p.v ? p.v + kvo : nullptr;
  bf16_t* const r_vt = p.vt ? p.vt + z * p.req.vt_bs : nullptr;

  if ((int)blockIdx.x < n_rope) {
    const int slot = tid % (16 * PC), hg = tid / (16 * PC);
    const int t = slot / PC, pc = slot - t * PC;
    const int s = blockIdx.x * 16 + t;
    if (hg >= HG || s >= p.S) return;
    const int d0 = pc * 8;
    float ca[8], sa[8], cb[8], sb[8];
    if (p.cosv) {
      const float* cr = p.cosv + (size_t)s * HD;
      const float* sr = p.sinv + (size_t)s * HD;
#pragma unroll
      for (int e = 0; e < 8; e += 4) {
        *(f32x4*)(ca + e) = *(const f32x4*)(cr + d0 + e);
        *(f32x4*)(sa + e) = *(const f32x4*)(sr + d0 + e);
        *(f32x4*)(cb + e) = *(const f32x4*)(cr + HALF + d0 + e);
        *(f32x4*)(sb + e) = *(const f32x4*)(sr + HALF + d0 + e);
      }
    } else {  // no rotary embedding (mllama vision tower, cross-attention q / k): pure head split
#pragma unroll
      for (int e = 0; e < 8; ++e) { ca[e] = 1.f; cb[e] = 1.f; sa[e] = 0.f; sb[e] = 0.f; }
    }
    const bf16_t* row = r_qkv + (size_t)s * p.ld_qkv;
    // four heads per round trip: their eight 16-byte loads are issued before the first is used (unconditional, clamped head
    // index; r01-r04 walked one head per iteration - one dependent load round trip per head: 11 of them in a row for the ViT's
    // 32 q / k heads on three head groups, 4.2 TB/s)
    constexpr int UNR = 4;
    const int nh = p.Hq + p.Hkv;
    for (int h0 = hg; h0 < nh; h0 += HG * UNR) {
      u32x4 ra[UNR], rb[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int head = min(h0 + u * HG, nh - 1);
        ra[u] = *(const u32x4*)(row + head * HD + d0);
        rb[u] = *(const u32x4*)(row + head * HD + HALF + d0);
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int head = h0 + u * HG;
        if (head >= nh) break;
        const bool is_q = head < p.Hq;
        const int hh = is_q ? head : head - p.Hq;
        float a[8], b[8], oa[8], ob[8];
        unpack8(ra[u], a);
        unpack8(rb[u], b);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          oa[e] = a[e] * ca[e] - b[e] * sa[e];  // first half: rotate_half gives -x2
          ob[e] = b[e] * cb[e] + a[e] * sb[e];  // second half: rotate_half gives +x1
        }
        bf16_t* dst = is_q ? r_q + ((size_t)hh * p.S + s) * HD
                           : r_k + ((size_t)hh * p.k_tokens + p.k_pos0 + s) * HD;
        *(u32x4*)(dst + d0) = pack8(oa);
        *(u32x4*)(dst + HALF + d0) = pack8(ob);
      }
    }
    return;
  }

  // ---- V block: (64 tokens, kv head hh)
  const int vb = blockIdx.x - n_rope;
  const int nblk = (p.S + 63) / 64;
  const int hh = vb / nblk;
  const int s0 = (vb - hh * nblk) * 64;
  const int head = p.Hq + p.Hkv + hh;
  const int ntok = min(64, p.S - s0);
  constexpr int VIT = (64 * CH + 255) / 256;   // 16-byte chunks per thread: 4 (head_dim 128) / 3 (80)
  u32x4 vraw[VIT];
#pragma unroll
  for (int i = 0; i < VIT; ++i) {               // every load of the thread before the first use (clamped: always a valid row)
    const int it = min(tid + i * 256, 64 * CH - 1);
    const int t = it / CH, c = it - t * CH;
    vraw[i] = *(const u32x4*)(r_qkv + (size_t)(s0 + min(t, ntok - 1)) * p.ld_qkv + head * HD + c * 8);
  }
#pragma unroll
  for (int i = 0; i < VIT; ++i) {
    const int it = tid + i * 256;
    if (it >= 64 * CH) break;
    const int t = it / CH, c = it - t * CH;
    u32x4 raw = (u32x4){0u, 0u, 0u, 0u};
    if (t < ntok) {
      const int s = s0 + t;
      raw = vraw[i];
      if (r_v) *(u32x4*)(r_v + ((size_t)hh * p.k_tokens + p.k_pos0 + s) * HD + c * 8) = raw;
    }
    if (r_vt) {
      const int tp = (t & 32) | (((t >> 2) & 3) << 3) | (((t >> 4) & 1) << 2) | (t & 3);   // key -> V^T column
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        tile[(c * 8 + 2 * e) * VT_LD + tp] = (bf16_t)(raw[e] & 0xffffu);
        tile[(c * 8 + 2 * e + 1) * VT_LD + tp] = (bf16_t)(raw[e] >> 16);
      }
    }
  }
  if (r_vt) {
    __syncthreads();
    for (int it = tid; it < HD * 8; it += 256) {
      const int d = it >> 3, c = it & 7;
      const u32x4 o = *(const u32x4*)(tile + d * VT_LD + c * 8);
      *(u32x4*)(r_vt + ((size_t)hh * HD + d) * p.vt_ld + s0 + c * 8) = o;
    }
  }
}

static int rope_split_launch(const RopeArgs& p, int HD, int nreq, hipStream_t stream) {
  const int n_rope = (p.S + 15) / 16;                                     // rope blocks: 16 tokens x all q / k heads
  const int n_v = (p.v || p.vt) ? ((p.S + 63) / 64) * p.Hkv : 0;          // V blocks: (64 tokens, kv head)
  const dim3 grid(n_rope + n_v, 1, nreq), block(256);
  vis_clear_error();
  if (HD == 128)
    hipLaunchKernelGGL(qkv_rope_split_kernel<128>, grid, block, 0, stream, p, n_rope);
  else
    hipLaunchKernelGGL(qkv_rope_split_kernel<80>, grid, block, 0, stream, p, n_rope);
  return vis_check_launch();
}

extern "C" int vis_qkv_rope_split(const void* qkv, const void* cosv, const void* sinv, void* q, void* k,
                                  void* v, void* vt, int S, int ld_qkv, int Hq, int Hkv, int HD,
                                  int k_tokens, int k_pos0, int vt_ld, hipStream_t stream) {
  // Hq == 0 (k/v only: cross-attention keys) and Hkv == 0 (q only) are allowed; cos == sin == NULL means no rotation
  if (!qkv || S <= 0 || Hq < 0 || Hkv < 0 || Hq + Hkv == 0) return VIS_ERR_ARG;
  if ((cosv == nullptr) != (sinv == nullptr)) return VIS_ERR_ARG;
  if ((Hq > 0 && !q) || (Hkv > 0 && !k)) return VIS_ERR_ARG;
  if (HD != 128 && HD != 80) return VIS_ERR_ARG;
  if (ld_qkv % 8 != 0 || ld_qkv < (Hq + 2 * Hkv) * HD) return VIS_ERR_ARG;
  if (Hkv > 0 && (k_pos0 < 0 || k_pos0 + S > k_tokens)) return VIS_ERR_ARG;
  if (vt && (vt_ld % 64 != 0 || vt_ld < ((S + 63) / 64) * 64)) return VIS_ERR_ARG;
  if (((uintptr_t)qkv | (uintptr_t)cosv | (uintptr_t)sinv | (uintptr_t)q | (uintptr_t)k | (uintptr_t)v |
       (uintptr_t)vt) & 15)
    return VIS_ERR_ARG;
  RopeArgs p;
  p.qkv = (const bf16_t*)qkv; p.cosv = (const float*)cosv; p.sinv = (const float*)sinv;
  p.q = (bf16_t*)q; p.k = (bf16_t*)k; p.v = (bf16_t*)v; p.vt = (bf16_t*)vt;
  p.S = S; p.ld_qkv = ld_qkv; p.Hq = Hq; p.Hkv = Hkv;
  p.k_tokens = k_tokens; p.k_pos0 = k_pos0; p.vt_ld = vt_ld;
  req_offsets_none(p.req);
  return rope_split_launch(p, HD, 1, stream);
}

// vis_qkv_rope_split for the `nreq` (<= 8) requests of a prompt-pass group in ONE launch: request r reads rows of qkv + r * qkv_bs
// (same S, same cos / sin rows: the requests of a group share their prompt structure) and writes q + r * q_bs, k + kv_off[r],
// v + kv_off[r] (element offsets of the request's cache slot from k / v; host array), vt + r * vt_bs.  Per request the same
// arithmetic as vis_qkv_rope_split; four launches of ~165 workgroups become one of ~660.
extern "C" int vis_qkv_rope_split_many(const void* qkv, const void* cosv, const void* sinv, void* q, void* k, void* v, void* vt,
                                       int S, int ld_qkv, int Hq, int Hkv, int HD, int k_tokens, int k_pos0, int vt_ld,
                                       int nreq, long long qkv_bs, long long q_bs, long long vt_bs, const long long* kv_off,
                                       hipStream_t stream) {
  if (!qkv || S <= 0 || Hq < 0 || Hkv < 0 || Hq + Hkv == 0 || nreq < 1 || nreq > VIS_MAX_REQ || !kv_off) return VIS_ERR_ARG;
  if ((cosv == nullptr) != (sinv == nullptr)) return VIS_ERR_ARG;
  if ((Hq > 0 && !q) || (Hkv > 0 && !k)) return VIS_ERR_ARG;
  if (HD != 128 && HD != 80) return VIS_ERR_ARG;
  if (ld_qkv % 8 != 0 || ld_qkv < (Hq + 2 * Hkv) * HD) return VIS_ERR_ARG;
  if (Hkv > 0 && (k_pos0 < 0 || k_pos0 + S > k_tokens)) return VIS_ERR_ARG;
  if (vt && (vt_ld % 64 != 0 || vt_ld < ((S + 63) / 64) * 64)) return VIS_ERR_ARG;
  if (((uintptr_t)qkv | (uintptr_t)cosv | (uintptr_t)sinv | (uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)vt) & 15)
    return VIS_ERR_ARG;
  if (qkv_bs < 0 || q_bs < 0 || vt_bs < 0 || (qkv_bs | q_bs | vt_bs) % 8) return VIS_ERR_ARG;
  RopeArgs p;
  p.qkv = (const bf16_t*)qkv; p.cosv = (const float*)cosv; p.sinv = (const float*)sinv;
  p.q = (bf16_t*)q; p.k = (bf16_t*)k; p.v = (bf16_t*)v; p.vt = (bf16_t*)vt;
  p.S = S; p.ld_qkv = ld_qkv; p.Hq = Hq; p.Hkv = Hkv;
  p.k_tokens = k_tokens; p.k_pos0 = k_pos0; p.vt_ld = vt_ld;
  req_offsets_none(p.req);
  for (int r = 0; r < nreq; ++r) {
    if (kv_off[r] < 0 || kv_off[r] % 8) return VIS_ERR_ARG;
    p.req.kv[r] = kv_off[r];
  }
  p.req.qkv_bs = qkv_bs; p.req.q_bs = q_bs; p.req.vt_bs = vt_bs;
  return rope_split_launch(p, HD, nreq, stream);
}

