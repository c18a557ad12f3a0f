/* vis_hip.h - C ABI of libvis_hip.so, the MI355X (gfx950) compute library behind
 * the Inspector/Auditor "image -> defect report" step.
 *
 * The reference (Aditya-Somasi/Vision-Inspection-System) has no FFI of its own:
 * its hot path is one remote call,
 *     client.chat.completions.create(model=, messages=, temperature=, max_tokens=)
 *     (src/agents/vlm_inspector.py:105-111, src/agents/vlm_auditor.py:117-129,:152-158)
 * and everything the remote service does for that call (image patch embedding,
 * ViT, merger, LLM prefill, greedy decode) is what these entry points compute
 * locally.  The arithmetic each entry point replaces is cited against the
 * model definition the service runs (TF = transformers/models/qwen2_vl).
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is a DEVICE pointer unless
 *    stated otherwise; bf16 tensors are raw uint16 bit patterns;
 *  - no allocation, no global state, no synchronisation: kernels are enqueued
 *    on `stream` (a hipStream_t passed as void*) and may be captured in a hipGraph;
 *  - return value: 0 = enqueued, 1 = argument/shape/alignment precondition
 *    violated (nothing launched), 2 = HIP launch error, 3 = valid arguments but
 *    this kernel form does not cover the shape on this device (nothing launched;
 *    the caller issues the equivalent separate launches - vis_decode_chain only).
 */
#ifndef VIS_HIP_H
#define VIS_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef void* vis_stream_t; /* hipStream_t */

#define VIS_OK 0
#define VIS_ERR_ARG 1
#define VIS_ERR_LAUNCH 2
#define VIS_ERR_UNSUPPORTED 3

/* activation selectors for vis_gemm_bf16 / vis_gemv_bf16 */
#define VIS_ACT_NONE 0
#define VIS_ACT_QUICKGELU 1 /* x * sigmoid(1.702 x)      ViT MLP (hidden_act="quick_gelu")   */
#define VIS_ACT_GELU_ERF 2  /* 0.5 x (1 + erf(x/sqrt2))  merger nn.GELU(), TF modeling:284   */
#define VIS_ACT_SWIGLU 3    /* silu(gate) * up over a 16-row interleaved gate/up weight       */

int vis_abi_version(void);

/* K2  C[M,N] = act(A[M,K] * W[N,K]^T + bias[N]) + R[M,N]      (bf16 in/out, f32 accumulate)
 * Replaces nn.Linear / Conv3d-as-GEMM in TF modeling_qwen2_vl.py:251-274 (patch embed),
 * :281-286 (merger), :296-298 (ViT MLP), :349-350,:421 (ViT qkv/proj), :459-466 (LLM MLP),
 * :501-504,:555 (LLM q/k/v/o).  K % 64 == 0, N % 4 == 0, lda/ldw % 8 == 0.
 * VIS_ACT_SWIGLU: W rows are [gate 16 | up 16] interleaved, output has N/2 columns. */
int vis_gemm_bf16(const void* A, const void* W, const void* bias, const void* R, void* C,
                  int M, int N, int K, int lda, int ldw, int ldc, int ldr, int act, vis_stream_t stream);

/* K3  y = w * bf16(x * rsqrt(mean(x^2) + eps))          TF modeling_qwen2_vl.py:96-110  */
int vis_rmsnorm_bf16(const void* x, const void* w, void* y, int rows, int N, int ldx, int ldy,
                     float eps, vis_stream_t stream);

/* K3 over the heads of a packed row: x [tokens][ldx] holds `heads` consecutive head_dim (= 128) wide heads per token; every
 * (token, head) slice is normalised with the one weight w [128] (mllama k_norm on the cross-attention keys, TF:models/mllama/
 * modeling_mllama.py:411-440).  Per slice bit-identical to vis_rmsnorm_bf16 on x[:, 128 h ..]; one launch instead of `heads`. */
int vis_rmsnorm_heads_bf16(const void* x, const void* w, void* y, int tokens, int heads, int head_dim, int ldx, int ldy,
                           float eps, vis_stream_t stream);

/* K5  y = (x - mean) * rsqrt(var + eps) * w + b         TF modeling_qwen2_vl.py:281,:428-429 */
int vis_layernorm_bf16(const void* x, const void* w, const void* b, void* y, int rows, int N,
                       int ldx, int ldy, float eps, vis_stream_t stream);

/* K4  rotary embedding + head split (+ KV-cache write, + V transpose).
 * qkv[S, (Hq+2Hkv)*HD] packed projections; cos/sin [S, HD] f32 rows (M-RoPE sections already
 * selected per channel, TF modeling_qwen2_vl.py:180-222; ViT 2-D rope :225-236).
 * q -> [Hq][S][HD]; k,v -> [Hkv][k_tokens][HD] rows k_pos0.. (v may be NULL);
 * vt -> [Hkv][HD][vt_ld] keys contiguous (may be NULL), in the k-slot column order vis_attn_prefill reads: inside
 * every aligned group of 32 keys, key 16a + 4h + r (a < 2, h < 4, r < 4) is column 8h + 4a + r (ABI version 2).
 * HD in {128, 80}.
 * Hq == 0 (k/v only) or Hkv == 0 (q only) give a partial split; cosv == sinv == NULL means no rotation (mllama
 * vision tower and cross-attention operands, TF:models/mllama/modeling_mllama.py:234-268,:411-440). */
int vis_qkv_rope_split(const void* qkv, const void* cosv, const void* sinv, void* q, void* k, void* v,
                       void* vt, int S, int ld_qkv, int Hq, int Hkv, int HD, int k_tokens, int k_pos0,
                       int vt_ld, vis_stream_t stream);

/* vis_qkv_rope_split for the nreq (<= 8) requests of one prompt-pass group as ONE launch (the engine's stacked suffix pass,
 * replacing the per-image loop of /root/reference/src/orchestration/graph.py:308-357 one level down): request r reads the S rows at
 * qkv + r * qkv_bs (same cos / sin rows for all: the requests share their prompt structure) and writes q + r * q_bs,
 * k + kv_off[r] and v + kv_off[r] (element offsets of its cache slot; HOST array of nreq values), vt + r * vt_bs.  Strides and
 * offsets in elements, multiples of 8.  Per request bit-identical to vis_qkv_rope_split. */
int vis_qkv_rope_split_many(const void* qkv, const void* cosv, const void* sinv, void* q, void* k, void* v, void* vt, int S,
                            int ld_qkv, int Hq, int Hkv, int HD, int k_tokens, int k_pos0, int vt_ld, int nreq,
                            long long qkv_bs, long long q_bs, long long vt_bs, const long long* kv_off, vis_stream_t stream);

/* K6/K7  flash-style prefill attention over a host-built work list of
 * {q0, qn<=128, k0, k1} int4 tiles (device memory): ViT varlen segments (cu_seqlens,
 * TF modeling_qwen2_vl.py:356-423) and LLM causal GQA prefill (:508-556).
 * O[Sq][ldo], column head*HD + d.  softmax(QK^T * scale) V, f32 softmax. */
int vis_attn_prefill(const void* Q, const void* K, const void* Vt, void* O, const void* work, int n_work,
                     int Hq, int Hkv, int HD, int Sq, int k_tokens, int vt_ld, int ldo, int causal,
                     float scale, vis_stream_t stream);
/* Same with a row offset: the work items' query rows (and the causal rule key <= query) are positions of the whole
 * sequence, Q / O hold only rows q_row0 .. q_row0 + Sq - 1 - the prompt pass of a request whose first q_row0 tokens
 * (the text prefix the reference puts in front of the image, src/agents/vlm_inspector.py:462-470, identical for the
 * images of a batch) were computed once and copied into its KV cache. */
int vis_attn_prefill_rows(const void* Q, const void* K, const void* Vt, void* O, const void* work, int n_work,
                          int Hq, int Hkv, int HD, int Sq, int k_tokens, int vt_ld, int ldo, int causal,
                          float scale, int q_row0, vis_stream_t stream);

/* vis_attn_prefill_rows (HD = 128) for the nreq (<= 8) requests of one prompt-pass group as ONE launch: request r reads Q + r * q_bs,
 * K + kv_off[r] (HOST array, element offsets), Vt + r * vt_bs and work items [r * n_work, (r + 1) * n_work) - one list per request
 * (the mllama cross-attention's key counts differ per image) - and writes O + r * o_bs.  Per request bit-identical. */
int vis_attn_prefill_rows_many(const void* Q, const void* K, const void* Vt, void* O, const void* work, int n_work, int Hq,
                               int Hkv, int HD, int Sq, int k_tokens, int vt_ld, int ldo, int causal, float scale, int q_row0,
                               int nreq, long long q_bs, long long vt_bs, long long o_bs, const long long* kv_off,
                               vis_stream_t stream);

/* K7, balanced form (HD = 128, causal, keys from 0): work = n_work x int4 {qB0, qBn, qA0, qAn}, a late and an early
 * 128-row query block of one sequence per 512-thread workgroup (qAn = 0: none), so that every workgroup of the causal
 * grid carries the same number of key tiles.  Results equal vis_attn_prefill_rows(..., causal = 1). */
int vis_attn_prefill_pairs(const void* Q, const void* K, const void* Vt, void* O, const void* work, int n_work,
                           int Hq, int Hkv, int HD, int Sq, int k_tokens, int vt_ld, int ldo, float scale,
                           int q_row0, vis_stream_t stream);

/* vis_attn_prefill_pairs for the nreq (<= 8) requests of one prompt-pass group as ONE launch (same work list): request r reads
 * Q + r * q_bs, K + kv_off[r] (HOST array, element offsets), Vt + r * vt_bs and writes O + r * o_bs.  Per request bit-identical to
 * vis_attn_prefill_pairs; the grid grows from Hq x n_work (168 workgroups for a 1289-row suffix) to Hq x n_work x nreq. */
int vis_attn_prefill_pairs_many(const void* Q, const void* K, const void* Vt, void* O, const void* work, int n_work, int Hq,
                                int Hkv, int HD, int Sq, int k_tokens, int vt_ld, int ldo, float scale, int q_row0, int nreq,
                                long long q_bs, long long vt_bs, long long o_bs, const long long* kv_off, vis_stream_t stream);

/* K6, key-split form (HD = 80, non-causal: the ViT towers).  Work items as in vis_attn_prefill plus SPLIT items, whose
 * y field is qn | (1 | part << 1 | pair << 2) << 8: parts 0 and 1 of pair `pair` cover the same query rows and the two
 * halves of their key range (cut at a multiple of 64); the kernel merges them (the later workgroup adds the earlier
 * one's partial, fixed operand order).  One 1024 x 1024 image at 16 heads leaves some SIMDs three 32-row blocks and
 * others two; with 9 of its 39 row blocks split every CU holds two whole and one half workgroup.
 * ws: vis_attn_split_ws_bytes(n_pairs, Hq) bytes, 256-byte aligned, zero before its first use (the kernel leaves it
 * reusable); one launch at a time per workspace.  Same reference as vis_attn_prefill (TF modeling_qwen2_vl.py:349-421). */
int vis_attn_split_ws_bytes(int n_pairs, int Hq);
int vis_attn_prefill_split(const void* Q, const void* K, const void* Vt, void* O, const void* work, int n_work,
                           int Hq, int Hkv, int HD, int Sq, int k_tokens, int vt_ld, int ldo, float scale,
                           int n_pairs, void* ws, long long ws_bytes, vis_stream_t stream);

/* K10  y[N] = act(W[N,K] x[K] + bias) + R, optional fused RMSNorm of x (norm_w != NULL).
 * out_f32 != 0 writes float logits (lm_head).  One generated token streams every weight once. */
int vis_gemv_bf16(const void* x, const void* W, const void* bias, const void* R, const void* norm_w,
                  void* y, int N, int K, int ldw, int act, int out_f32, float eps, vis_stream_t stream);

/* K4 + K11 (decode)  one launch per layer for the single new token: M-RoPE of q,k from the packed
 * projection row (cos/sin row *step_ptr of the [cache_tokens][128] f32 tables), KV-cache append at slot
 * *step_ptr, GQA attention over the *step_ptr + 1 cached keys split nsplit ways over the context
 * (nsplit * 128 >= cache_tokens), then a combine launch.  step_ptr is a device int: graph-replayable.
 * Workspaces: part_o [Hq][nsplit][128] f32, part_ml [Hq][nsplit][2] f32.  out: [Hq*128] bf16.
 * Replaces TF modeling_qwen2_vl.py:180-222 + :508-556 for q_len == 1 with a KV cache. */
int vis_decode_attn(const void* qkv, const void* cos_t, const void* sin_t, void* k_cache, void* v_cache,
                    const void* step_ptr, void* part_o, void* part_ml, void* out, int Hq, int Hkv, int HD,
                    int cache_tokens, int nsplit, float scale, int batch, long long qkv_bs, long long cache_bs,
                    long long tab_bs, vis_stream_t stream);
/* batch > 1: sequence b uses qkv + b*qkv_bs, caches + b*cache_bs, tables + b*tab_bs (element strides),
 * step_ptr[b], part_o/part_ml/out blocks of Hq*nsplit*128 / Hq*nsplit*2 / Hq*128 elements. */

/* vis_decode_attn for a batch whose sequences share their first shared_len cached keys (a multiple of 64: the common text
 * prefix of a batch inspection - the reference sends the text part first, src/agents/vlm_inspector.py:462-470 - which the
 * prompt passes copied into every slot): every sequence reads those keys / values from sequence 0's cache (one HBM read +
 * L2 hits instead of `batch` HBM reads of identical rows).  Bit-identical to vis_decode_attn. */
int vis_decode_attn_shared(const void* qkv, const void* cos_t, const void* sin_t, void* k_cache, void* v_cache,
                           const void* step_ptr, void* part_o, void* part_ml, void* out, int Hq, int Hkv, int HD,
                           int cache_tokens, int nsplit, float scale, int batch, long long qkv_bs, long long cache_bs,
                           long long tab_bs, int shared_len, vis_stream_t stream);

/* vis_decode_attn_shared with the qkv row finalised inside the attention launch: `part` = the ksplit f32 partial slabs
 * [slab_rows][(Hq + 2 Hkv) * 128] the batched qkv projection (vis_gemm_decode_bf16 / _fp8) left (slab_rows = 16 / 32 / 64 for batch
 * <= 16 / <= 32 / beyond), `bias` [n] or NULL, `sx` [batch] / `sw` [n] the e4m3 scales of fp8 partials or both NULL.  Column n of
 * sequence b = bf16(sum_k part[k][b][n] (* sx[b] * sw[n]) + bias[n]): vis_skinny_finalize's arithmetic bit for bit, i.e. the pair
 * (vis_skinny_finalize[_fp8], vis_decode_attn_shared) as ONE launch.  shared_len as above (0: none). */
int vis_decode_attn_parts(const void* part, int ksplit, int slab_rows, const void* bias, const void* sx, const void* sw,
                          const void* cos_t, const void* sin_t, void* k_cache, void* v_cache, const void* step_ptr,
                          void* part_o, void* part_ml, void* out, int Hq, int Hkv, int HD, int cache_tokens, int nsplit,
                          float scale, int batch, long long cache_bs, long long tab_bs, int shared_len, vis_stream_t stream);

/* K10 + K4 + K11 + K10 (single-sequence decode)  the head of a decoder layer as ONE launch:
 *   qkv = W_qkv rmsnorm(x) + b ; attn = attention(rope(q), cache + rope(k), v) ; y = x + W_o attn
 * i.e. vis_gemv_bf16 (norm fused) + vis_decode_attn (split + combine launches) + vis_gemv_bf16 (residual), bit-identical to
 * those four launches (csrc/decode_chain.hip: the stages are workgroup roles of one resident grid that hand their results
 * over as tagged 8-byte granules).  Replaces TF modeling_qwen2_vl.py:501-556 (+ :96-110 input norm) for q_len == 1 with a
 * KV cache.  x [K] bf16 layer input - or, with x_idx != NULL, an [x_rows][K] table whose row *x_idx (a device int, clamped
 * like vis_gather_rows) is the input: the first layer reads the new token's embedding row itself -, Wqkv [(Hq + 2 Hkv) * 128][ldw_qkv], bqkv or NULL, norm_w [K], Wo [K][ldw_o], y [K];
 * tables, caches, step_ptr as vis_decode_attn.
 * ws: vis_decode_chain_ws_bytes(Hq, Hkv, nsplit) bytes and sync: vis_decode_chain_sync_ints() ints, both zeroed once by the
 * caller and owned by one stream; sync[0] counts completed launches, sync[32] is a status word: non-zero after a launch = a
 * bounded wait gave up (results invalid; zero ws and sync before the next launch - until then every further launch on this
 * sync block returns at once without computing: a stranded request costs one wait bound, not one per launch).  The tag of a
 * launch is sync[0] + 1, and 1 after 2^32 - 1 (tag 0 marks a never-written granule): zero ws and sync between requests
 * before sync[0] gets near 2^32.  The packed projection row and the merged
 * attention row live in ws as granules (low 32 bits = two bf16): ws[0 .. (Hq + 2 Hkv) * 64) and the next Hq * 64 words.
 * VIS_ERR_UNSUPPORTED for shapes outside the chained form (HD != 128, Hq / Hkv not in {1, 2, 4, 7, 8}, K or Hq * 128 > 4096,
 * Hq > 64, or more WAITING workgroups - the projection and merge roles plus Hkv attention items per 64 keys of ctx_bound -
 * than the device holds resident minus a margin of 32): the caller then issues the four launches.  ctx_bound: the largest
 * context (cached keys incl. the new one) any launch with these arguments will see (a captured launch is replayed at growing
 * positions), <= 0 = cache_tokens; attention items of splits past the context return at once and need no residency.
 * VIS_ERR_ARG is a caller bug.  ONE chained launch at a time per device: its workgroups wait
 * for each other inside the grid, so launches from two streams at once must be ordered by the caller (an event). */
int vis_decode_chain_sync_ints(void);
long long vis_decode_chain_ws_bytes(int Hq, int Hkv, int nsplit);
/* largest ctx_bound vis_decode_chain accepts for (Hq, Hkv, head_dim 128, hidden K) on the current device; 0 = shape not covered */
int vis_decode_chain_ctx_limit(int Hq, int Hkv, int K);
int vis_decode_chain(const void* x, const void* x_idx, int x_rows, const void* Wqkv, const void* bqkv, const void* norm_w, const void* Wo, void* y,
                     const void* cos_t, const void* sin_t, void* k_cache, void* v_cache, const void* step_ptr,
                     void* ws, void* sync, int Hq, int Hkv, int HD, int K, int ldw_qkv, int ldw_o,
                     int cache_tokens, int nsplit, int ctx_bound, float scale, float eps, vis_stream_t stream);

/* K10 + K12  lm_head of the single-sequence step with the pick's first stage in its epilogue, then the merging launch:
 * logits[N] f32 = W rmsnorm(x); tokens[*step] = cur_token = pick; *step += 1 - the pick vis_gemv_bf16 + vis_argmax_f32 make
 * (same comparison, same Gumbel noise), one launch fewer.  ws_val / ws_idx: 2048 floats / ints. */
int vis_gemv_bf16_argmax(const void* x, const void* W, const void* norm_w, void* logits, int N, int K, int ldw, float eps,
                         void* ws_val, void* ws_idx, void* tokens, int max_tokens, void* cur_token, void* step_ptr,
                         float inv_temp, unsigned seed, vis_stream_t stream);

/* K12  next-token pick: tokens[*step] = cur_token = argmax(logits) (first index on ties, like
 * torch.argmax), then *step += 1.  inv_temp > 0 samples at temperature 1/inv_temp by Gumbel-max with a
 * counter hash of (seed, *step, index); inv_temp == 0 is greedy (the reference request passes
 * temperature=, src/agents/vlm_inspector.py:108).  ws_val/ws_idx: 256 floats / 256 ints of workspace. */
int vis_argmax_f32(const void* logits, int V, void* ws_val, void* ws_idx, void* tokens, int max_tokens,
                   void* cur_token, void* step_ptr, float inv_temp, unsigned seed, int batch, int ld_logits,
                   vis_stream_t stream);
/* batch > 1: sequence b reads logits + b*ld_logits, writes tokens[b*max_tokens + step[b]], cur_token[b],
 * step_ptr[b]; ws_val / ws_idx need 256 entries per sequence. */

/* K10 (batched decode), first half.  For up to 64 in-flight sequences the weight matrix is streamed from HBM
 * ONCE per step by <= 256 persistent workgroups (one per CU, 7-stage LDS-DMA ring, stream-K cut of the
 * (128-column tile, K-step) sequence).  part[slot][R][N] (f32), R = 16 / 32 / 64 for B <= 16 / 32 / 64 (one, two or four
 * 16-row MFMA blocks; vis_skinny_finalize uses the same rule): `ksplit` slots, ALL written (unused ones
 * zero-filled), sum over slots = x[B,K] * W[N,K]^T; the slot order is fixed by (N, K) alone, so results are
 * bitwise reproducible.  ksplit <= 0 means vis_gemm_decode_ksplit(N, K) = the slots the geometry needs (<= 16);
 * a smaller value is an argument error.  part == NULL: C is written directly (bf16, or f32 logits when out_f32). */
int vis_gemm_decode_ksplit(int N, int K);
int vis_gemm_decode_bf16(const void* x, const void* W, void* part, void* C, int B, int N, int K, int ldx, int ldw,
                         int ldc, int ksplit, int out_f32, vis_stream_t stream);

/* K10 (batched decode), second half: one workgroup per sequence sums the K-slices in a fixed order (bitwise
 * reproducible), applies bias / residual or SwiGLU (16-column interleaved gate/up, N/2 outputs) -> y, and when
 * norm_w != NULL also writes yn = bf16(bf16(y * rsqrt(mean(y^2) + eps)) * norm_w) - the RMSNorm'd input of the
 * NEXT projection (TF modeling_qwen2_vl.py:96-110,:459-466,:598-624). */
int vis_skinny_finalize(const void* part, int ksplit, const void* bias, const void* R, const void* norm_w, void* y,
                        void* yn, int B, int N, int ldr, int ldy, int ldyn, int swiglu, float eps,
                        vis_stream_t stream);

/* Row f3: bicubic resample of the decoded RGB frame, bit-exact with Pillow's 8-bit ImagingResample (the
 * resampler behind the HF Qwen2-VL processor's resize(resample=BICUBIC),
 * TF:models/qwen2_vl/image_processing_qwen2_vl.py:62-89,:165-198; replaces the host-side PIL call of the service
 * half of reference step a3, src/agents/vlm_inspector.py:46-88).  src u8 [in_h][in_w][3] -> dst u8
 * [out_h][out_w][3], tmp u8 [in_h][out_w][3].  kx int32 [out_w][ksx] / bx int32 [out_w][2] = {first column,
 * taps}: fixed-point (22 fractional bits) filter rows of the horizontal pass; ky/by the same for the vertical pass
 * (built on the host by image_processing.resample_coeffs).  All pointers are device pointers. */
int vis_resize_rgb_u8(const void* src, void* tmp, void* dst, int in_h, int in_w, int out_h, int out_w,
                      const void* kx, const void* bx, int ksx, const void* ky, const void* by, int ksy, void* stream);

/* Service-side JPEG decode, GPU half (csrc/jpeg.hip; host half: include/vis_jpeg_host.h).  Replaces the libjpeg decode
 * of the data-URI image the reference's agents send (src/agents/vlm_inspector.py:46-88 writes it; the service reads it).
 * coeffs: int16 [blocks][64] quantised DCT coefficients in natural order, component planes back to back (Y, Cb, Cr),
 * blocks row-major; qt: int32 [3][64]; planes: workspace of blocks * 64 bytes; rgb: uint8 [height][width][3].
 * hs / vs: luma sampling factors (1x1, 2x1 or 2x2; chroma is 1x1); bw / bh: blocks per row / column of the luma and the
 * chroma planes; dw_c / dh_c: real chroma size in samples.  Integer arithmetic throughout: the result equals
 * libjpeg-turbo's default decoder (islow IDCT, fancy upsampling) bit for bit. */
int vis_jpeg_to_rgb(const void* coeffs, const void* qt, void* planes, void* rgb, int width, int height, int ncomp,
                    int hs, int vs, int bw_y, int bh_y, int bw_c, int bh_c, int dw_c, int dh_c, vis_stream_t stream);

/* K1 (front)  resized RGB u8 frame [H][W][3] -> normalised bf16 patch rows
 * out[row0 + p][ld_out] in the merge-group order of
 * TF image_processing_pil_qwen2_vl.py:156-190; mean/stdv are HOST pointers to 3 floats. */
int vis_patchify_u8(const void* img, void* out, int H, int W, int ld_out, int row0, const float* mean,
                    const float* stdv, vis_stream_t stream);

/* K12  out[i][:] = table[ids[i]][:] (embedding lookup, ids int32 on device). */
int vis_gather_rows(const void* table, const void* ids, void* out, int n, int D, int n_table,
                    vis_stream_t stream);

/* K12  dst[idx[i]][:] = src[i][:] (image-token scatter, TF modeling_qwen2_vl.py:1144-1200). */
int vis_scatter_rows(const void* src, const void* idx, void* dst, int n, int D, int n_dst,
                     vis_stream_t stream);

/* K2, split-K form for long-K problems whose 256x256 tile count cannot fill the chip (LLM down projection: 126 tiles,
 * K = 18944): `ksplit` (2..8) K-slices run as independent tiles, f32 partials go to `work` (ksplit*M*N floats), a
 * second launch sums them in a fixed order and applies bias / act (0..2) / residual.  N % 8 == 0, no SwiGLU. */
int vis_gemm_bf16_splitk(const void* A, const void* W, const void* bias, const void* R, void* C, void* work, int M, int N,
                         int K, int lda, int ldw, int ldc, int ldr, int act, int ksplit, vis_stream_t stream);
/* The K-sliced tiles of vis_gemm_bf16_splitk alone: work[ksplit][M][N] f32 partial sums of A W^T (no finalisation). */
int vis_gemm_bf16_splitk_part(const void* A, const void* W, void* work, int M, int N, int K, int lda, int ldw,
                              int ksplit, vis_stream_t stream);
/* Their finalisation fused with the NEXT norm of the block (K3 / K5; the row owner exists here):
 * x = bf16(sum_s part[s] + bias + R); y = RMSNorm(x) * norm_w (norm_b == NULL) or LayerNorm(x) (norm_b != NULL), with
 * the statistics of the rounded x - bit-identical to vis_gemm_bf16_splitk followed by vis_rmsnorm_bf16 /
 * vis_layernorm_bf16.  y == NULL: finalisation only.  N <= 5120, N % 8 == 0. */
int vis_splitk_finalize_norm(const void* part, int ksplit, const void* bias, const void* R, void* x,
                             const void* norm_w, const void* norm_b, void* y, int M, int N, int ldr, int ldx, int ldy,
                             float eps, vis_stream_t stream);

/* BASELINE configs[4] slice - fp8 weights for the HBM-bound decode projections ("W8A16"):
 * y = act((Wq x) * scale + bias) + R with Wq OCP e4m3 bytes [N][ldw] and a per-output-row f32 scale; x bf16 with the
 * same fused RMSNorm prologue / bias / residual / SwiGLU (16-row interleaved gate/up) / f32-output options as
 * vis_gemv_bf16.  Halves the bytes per generated token (14.14 GB -> 7.07 GB at 7B). */
int vis_gemv_fp8w(const void* x, const void* Wq, const void* scale, const void* bias, const void* R,
                  const void* norm_w, void* y, int N, int K, int ldw, int act, int out_f32, float eps,
                  vis_stream_t stream);

/* K10 (batched decode), round 5: projection + split-K reduction + row-wise epilogue in ONE launch (csrc/decode_stream.hip);
 * replaces vis_gemm_decode_* + vis_skinny_finalize* in the engines (those stay exported).  Same persistent stream-K ring;
 * a tile cut between workgroups is summed, in fixed segment order, by whichever of them takes the tile's last ticket
 * (nobody waits), then finished there:
 *   VIS_DP_PLAIN       C[b][n] = (x W^T)[b][n] * rs[b] + bias[n]         bf16, or f32 when out_f32 (q/k/v, lm_head)
 *   VIS_DP_SWIGLU      C[b][o] = silu(g * rs[b]) * (u * rs[b])           16-row interleaved gate/up weight, N / 2 outputs
 *   VIS_DP_RESID_NORMW C = y = bf16(x W^T + R), Cw = bf16(y * nw[n]), ssq_out[n / 32][b] = sum over the 32-column unit of y^2
 * The RMSNorm in front of a projection (TF modeling_qwen2_vl.py:96-110) is applied in two exact halves: the PRODUCER of the
 * row multiplies by the norm weight (Cw, a per-column factor), the CONSUMER by rs[b] = rsqrt(sum_t ssq_in[t][b] / norm_dim +
 * eps) (a per-row factor that commutes with the projection); ssq_in == NULL means rs = 1.  ssq buffers: [columns / 32][64]
 * f32 (tiles_in = norm_dim / 32 <= 128 units).
 * ws: vis_decode_proj_ws_bytes(B, N, K, fp8) bytes, 256-byte aligned, zeroed once (the kernel leaves it reusable), one
 * launch at a time per workspace.  A row's result depends on that row alone (bitwise slot / batch-size invariance).
 * Cq / Cqs (SWIGLU, RESID_NORMW; may be NULL): the row the next projection consumes (act, or y * nw) as MX blocks: OCP
 * e4m3 bytes [B][ldcq] + one E8M0 scale byte per 32 columns [B][ldcqs] (the smallest power of two with block maximum /
 * scale <= 448).  vis_decode_proj_fp8 (BASELINE configs[4]): A given as such blocks (Aq, As), Wq e4m3 [N][ldw] with
 * per-output-row f32 scales sw[N], on v_mfma_scale_f32_16x16x128_f8f6f4 with the block scale in the instruction's scale
 * operand; K % 128 == 0.  vis_decode_prep_rows: head of a step - x[b] = table[ids[b]] (clamped), xw = bf16(x * nw), the
 * MX copy of xw (xq / xqs, may be NULL) and ssq[n / 32][b]; H % 128 == 0. */
#define VIS_DP_PLAIN 0
#define VIS_DP_SWIGLU 1
#define VIS_DP_RESID_NORMW 2
long long vis_decode_proj_ws_bytes(int B, int N, int K, int fp8);
int vis_decode_proj_bf16(const void* A, const void* W, void* ws, void* C, void* Cw, void* Cq, void* Cqs,
                         const void* bias, const void* R, const void* nw, const void* ssq_in, void* ssq_out,
                         int B, int N, int K, int lda, int ldw, int ldc, int ldr, int ldcq, int ldcqs, int mode,
                         int out_f32, int tiles_in, int norm_dim, float eps, vis_stream_t stream);
int vis_decode_proj_fp8(const void* Aq, const void* As, const void* Wq, const void* sw, void* ws, void* C,
                        void* Cw, void* Cq, void* Cqs, const void* bias, const void* R, const void* nw,
                        const void* ssq_in, void* ssq_out, int B, int N, int K, int ldaq, int ldas, int ldw,
                        int ldc, int ldr, int ldcq, int ldcqs, int mode, int out_f32, int tiles_in,
                        int norm_dim, float eps, vis_stream_t stream);
/* Column-parallel form of the same projection (csrc/decode_colpar.hip): every workgroup owns whole output columns - floor or
 * ceil((N / 32) / workgroups) units of 32 columns, at most five, min(256, N / 32) workgroups - and the entire K, so nothing is
 * reduced across workgroups: no workspace, no tickets, the epilogue runs on registers.  Each workgroup streams all of x
 * (rows x K, from L2) next to its weight rows (from HBM): the form for projections whose x is small next to the weights a
 * workgroup owns (everything but the long-K down projection at many sequences).  Arguments, epilogue modes and per-row
 * results as vis_decode_proj_* (a row's sum runs over the K-steps in ascending order in both forms; tiles the stream-K form
 * does not cut are bit-identical).  vis_decode_proj_colpar_covers: 1 when the form covers (N, mode, MX output wanted) - N %
 * 32 == 0, N <= 40960, not SwiGLU with an MX output (an act block of 32 columns spans two units); otherwise the entry
 * points return VIS_ERR_UNSUPPORTED and the caller uses vis_decode_proj_*. */
int vis_decode_proj_colpar_covers(int N, int mode, int mx_out);
int vis_decode_proj_colpar_bf16(const void* A, const void* W, void* C, void* Cw, void* Cq, void* Cqs, const void* bias,
                                const void* R, const void* nw, const void* ssq_in, void* ssq_out, int B, int N, int K,
                                int lda, int ldw, int ldc, int ldr, int ldcq, int ldcqs, int mode, int out_f32,
                                int tiles_in, int norm_dim, float eps, vis_stream_t stream);
int vis_decode_proj_colpar_fp8(const void* Aq, const void* As, const void* Wq, const void* sw, void* C, void* Cw,
                               void* Cq, void* Cqs, const void* bias, const void* R, const void* nw,
                               const void* ssq_in, void* ssq_out, int B, int N, int K, int ldaq, int ldas, int ldw,
                               int ldc, int ldr, int ldcq, int ldcqs, int mode, int out_f32, int tiles_in,
                               int norm_dim, float eps, vis_stream_t stream);
int vis_decode_prep_rows(const void* table, const void* ids, const void* nw, void* x, void* xw, void* xq,
                         void* xqs, void* ssq, int B, int table_rows, int H, int ldx, int ldq, int ldqs,
                         vis_stream_t stream);

/* K10 for a handful of in-flight sequences (1 <= B <= 4): vis_gemv_bf16 / vis_gemv_fp8w over B input rows
 * (x [B][ldx], y [B][ldy], R [B][ldr]; bias and norm_w shared).  The weights are streamed ONCE for all rows, each row's
 * arithmetic is the single-row kernel's (bit-identical results), and - unlike vis_gemm_decode_* + vis_skinny_finalize* -
 * there is no partial buffer and no finalisation launch.  B * K * 2 bytes of LDS (<= 152 KiB). */
int vis_gemv_bf16_rows(const void* x, const void* W, const void* bias, const void* R, const void* norm_w, void* y,
                       int B, int N, int K, int ldw, int ldx, int ldy, int ldr, int act, int out_f32, float eps,
                       vis_stream_t stream);
int vis_gemv_fp8w_rows(const void* x, const void* Wq, const void* scale, const void* bias, const void* R,
                       const void* norm_w, void* y, int B, int N, int K, int ldw, int ldx, int ldy, int ldr, int act,
                       int out_f32, float eps, vis_stream_t stream);

/* BASELINE configs[4]: GEMM on the CDNA4 block-scaled fp8 MFMA (v_mfma_scale_f32_16x16x128_f8f6f4, unit block scales).
 * C[M, N(/2)] = act((Aq Wq^T) * sa[m] * sw[n] + bias) + R with Aq [M][lda] / Wq [N][ldw] OCP e4m3 bytes, per-row f32
 * scales sa [M] (per token, vis_quant_rows_fp8) and sw [N] (per output channel); act as vis_gemm_bf16; K % 128 == 0.
 * work != NULL: split-K over `ksplit` (2..8) K-slices with f32 partials in work (ksplit*M*N floats) and a fixed-order
 * finalisation (long-K problems with too few 256x256 tiles; N % 8 == 0, no SwiGLU); work == NULL: single pass. */
int vis_gemm_fp8(const void* Aq, const void* sa, const void* Wq, const void* sw, const void* bias, const void* R,
                 void* C, void* work, int ksplit, int M, int N, int K, int lda, int ldw, int ldc, int ldr, int act,
                 vis_stream_t stream);

/* Per-row dynamic quantisation of bf16 activations to e4m3: scale[m] = amax(row)/448, q = rne(x * (1/scale)); with
 * norm_w != NULL the row is normalised first (K <= 4096): RMSNorm when norm_b == NULL, LayerNorm otherwise - the
 * same bf16 values vis_rmsnorm_bf16 / vis_layernorm_bf16 write. */
int vis_quant_rows_fp8(const void* x, const void* norm_w, const void* norm_b, void* q, void* scale, int rows, int K,
                       int ldx, int ldq, float eps, vis_stream_t stream);

/* BASELINE configs[4], batched decode: fp8 forms of vis_gemm_decode_bf16 / vis_skinny_finalize.
 * vis_gemm_decode_fp8: xq [B][ldx] / Wq [N][ldw] OCP e4m3 bytes (K % 128 == 0) on the block-scaled fp8 MFMA, same
 * persistent stream-K ring; partials (part != NULL) are RAW sums, the direct form (part == NULL) applies sx[b]*sw[n].
 * vis_skinny_finalize_fp8: sums the partial slots, scales by sx[b]*sw[n], then bias / residual / SwiGLU / next
 * RMSNorm as vis_skinny_finalize; yq != NULL additionally writes the row the NEXT projection consumes (yn if normed,
 * else y) as e4m3 bytes with its per-row scale yq_scale[b] = amax/448 (the activation quantiser, fused). */
int vis_gemm_decode_fp8_ksplit(int N, int K);
int vis_gemm_decode_fp8(const void* xq, const void* sx, const void* Wq, const void* sw, void* part, void* C, int B,
                        int N, int K, int ldx, int ldw, int ldc, int ksplit, int out_f32, vis_stream_t stream);
int vis_skinny_finalize_fp8(const void* part, int ksplit, const void* sx, const void* sw, const void* bias,
                            const void* R, const void* norm_w, void* y, void* yn, void* yq, void* yq_scale, int B,
                            int N, int ldr, int ldy, int ldyn, int ldyq, int swiglu, float eps, vis_stream_t stream);

/* ---- Row f2: Llama-3.2-11B-Vision ("mllama") Auditor, reference src/agents/vlm_auditor.py:81-83,:152-158 ---- */

/* Resized RGB frame -> patch rows of the tile canvas (zero padding applied to RAW pixels, then rescale/normalise;
 * TF:models/mllama/image_processing_pil_mllama.py pad/split_to_tiles_np; Conv2d feature order (c, ph, pw),
 * TF:models/mllama/modeling_mllama.py:833-840).  Row of tile t, patch (py, px): t*((tile/14)^2+1) + 1 + py*(tile/14) + px;
 * CLS rows and absent tiles are not written (the caller zero-fills). */
int vis_patchify_tiles_u8(const void* img, void* out, int H, int W, int tiles_h, int tiles_w, int tile, int ld_out,
                          const float* mean, const float* stdv, vis_stream_t stream);

/* x[i][:] += table[idx[i]][:] (bf16, f32 add): per-tile embeddings broadcast over a tile's tokens
 * (TF:models/mllama/modeling_mllama.py:102-122). */
int vis_add_rows_bf16(void* x, const void* table, const void* idx, int n, int D, int ldx, int n_table,
                      vis_stream_t stream);

/* Decode-step cross-attention over a static key/value set (TF:models/mllama/modeling_mllama.py:384-466): q [Hq*128]
 * from the q projection (per-head q_norm applied inside), k/v [Hkv][key_tokens][128], *nkeys_m1 = valid keys - 1. */
int vis_decode_cross_attn(const void* q, const void* q_norm_w, const void* k, const void* v, const void* nkeys_m1,
                          void* part_o, void* part_ml, void* out, int Hq, int Hkv, int HD, int key_tokens,
                          int nsplit, float scale, float eps, vis_stream_t stream);
/* Batch form: sequence b uses q + b * q_bs, k / v + b * kv_bs (element strides) and nkeys_m1[b]; out [batch][Hq*128]. */
int vis_decode_cross_attn_batch(const void* q, const void* q_norm_w, const void* k, const void* v,
                                const void* nkeys_m1, void* part_o, void* part_ml, void* out, int Hq, int Hkv, int HD,
                                int key_tokens, int nsplit, float scale, float eps, int batch, long long q_bs,
                                long long kv_bs, vis_stream_t stream);

/* Row f4: pixel statistics of the image-quality pre-check (src/safety/image_quality.py:42-56,:118-127): exact integer
 * sums over the RGB frame - stats[0] = sum gray, stats[1] = sum Laplacian, stats[2] = sum Laplacian^2 (int64[3]) with
 * OpenCV's published 8-bit RGB2GRAY fixed-point rule and the ksize-1 Laplacian under BORDER_REFLECT_101.
 * Parity with the reference's cv2 calls is UNPINNED (OpenCV is absent here; the reference holds no fixtures). */
int vis_image_stats_u8(const void* img, int H, int W, void* stats, vis_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VIS_HIP_H */
