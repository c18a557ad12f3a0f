"""Qwen2-VL inference engine on MI355X: image frames + token ids -> generated token ids.

This is the local replacement of what the reference's remote endpoint does for one
``chat.completions.create`` call (src/agents/vlm_inspector.py:105-111).  The control flow is
Python; every arithmetic op is a gfx950 HIP kernel reached through ``hip.py`` (C ABI in
include/vis_hip.h).  PyTorch only allocates device memory, provides the HIP stream and
captures the per-token decode step into a hipGraph (``torch.cuda.CUDAGraph``).

Stages (kernel ids as in SURVEY.md section 8a):
  vision : K1 patchify+GEMM, then 32 x [K5 LN, K2 qkv GEMM, K4 2-D rope/split, K6 varlen attention,
           K2 proj(+res), K5 LN, K2 fc1(+QuickGELU), K2 fc2(+res)], merger (K5, K2+GELU, K2)
  prefill: K12 embed gather + image scatter, 28 x [K3, K2 qkv(+bias), K4 M-RoPE/split/KV write,
           K7 causal GQA attention, K2 o(+res), K3, K2 gate/up(+SwiGLU), K2 down(+res)], K10 lm_head, K12 argmax
  decode : per token, one hipGraph replay of 28 x [K10 qkv(+RMSNorm,+bias), K4 rope+KV append,
           K11 attention, K10 o(+res), K10 gate/up(+RMSNorm,+SwiGLU), K10 down(+res)], K10 lm_head, K12 argmax
"""
from __future__ import annotations

import collections
import logging
import os
import threading
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import hip
from .config import Qwen2VLConfig
from .weights import DeviceWeights, PATCH_K_PAD


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


# ----------------------------------------------------------------------------- host-side position math
def vision_pos_hw(grids: Sequence[Tuple[int, int, int]], merge: int) -> np.ndarray:
    """(h, w) index of every patch, 2x2-merge-block order (TF vision_utils.get_vision_position_ids)."""
    out = []
    for (t, h, w) in grids:
        hp = np.broadcast_to(np.arange(h)[:, None], (h, w))
        wp = np.broadcast_to(np.arange(w)[None, :], (h, w))
        shape = (h // merge, merge, w // merge, merge)
        hp = hp.reshape(shape).transpose(0, 2, 1, 3).reshape(-1)
        wp = wp.reshape(shape).transpose(0, 2, 1, 3).reshape(-1)
        out.append(np.tile(np.stack([hp, wp], axis=-1), (t, 1)))
    return np.concatenate(out, axis=0)


def vision_cos_sin(cfg: Qwen2VLConfig, grids) -> Tuple[np.ndarray, np.ndarray]:
    """cos/sin rows [N, 80] f32 of the ViT 2-D rotary embedding (TF modeling_qwen2_vl.py:239-248,:719-722)."""
    dim = cfg.v_head_dim // 2
    inv_freq = (1.0 / (10000.0 ** (np.arange(0, dim, 2, dtype=np.float32) / np.float32(dim)))).astype(np.float32)
    pos = vision_pos_hw(grids, cfg.merge).astype(np.float32)
    freqs = (pos[:, :, None] * inv_freq[None, None, :]).reshape(pos.shape[0], -1)
    emb = np.concatenate([freqs, freqs], axis=-1)
    return np.cos(emb).astype(np.float32), np.sin(emb).astype(np.float32)


def vision_window_order(cfg: Qwen2VLConfig, grid: Tuple[int, int, int]) -> Tuple[np.ndarray, List[int]]:
    """Qwen2.5-VL window layout of ONE image (TF vision_utils.get_vision_window_index): the permutation of its 2x2 merge
    units into window order and the cumulative window boundaries in patch rows.  Windows are v_window pixels square
    (v_window / merge / patch merged tokens per side); the right / bottom windows of a grid are ragged."""
    t, h, w = grid
    win = cfg.v_window // cfg.merge // cfg.patch
    unit = cfg.merge ** 2
    lh, lw = h // cfg.merge, w // cfg.merge
    idx = np.arange(t * lh * lw).reshape(t, lh, lw)
    ph, pw = win - lh % win, win - lw % win               # the reference pads a whole window when already divisible
    nh, nw = (lh + ph) // win, (lw + pw) // win
    pad = np.pad(idx, ((0, 0), (0, ph), (0, pw)), constant_values=-100)
    pad = pad.reshape(t, nh, win, nw, win).transpose(0, 1, 3, 2, 4).reshape(t, nh * nw, win * win)
    seqlens = (pad != -100).sum(axis=2).reshape(-1)
    flat = pad.reshape(-1)
    cu = [0]
    for n in seqlens.tolist():
        if n:                                             # all-padding windows vanish (unique_consecutive)
            cu.append(cu[-1] + n * unit)
    return flat[flat != -100].astype(np.int64), cu


def rope_index(cfg: Qwen2VLConfig, input_ids: Sequence[int], grids) -> Tuple[np.ndarray, int]:
    """M-RoPE position ids [3, S] of one sequence and the next text position
    (TF modeling_qwen2_vl.py:914-1016: text runs count up on all 3 axes, an image block uses
    (t, h, w) over the merged grid and advances the text position by max(h, w) / merge)."""
    ids = np.asarray(list(input_ids))
    is_img = ids == cfg.image_token_id
    pos: List[np.ndarray] = []
    cur, i, g, n = 0, 0, 0, len(ids)
    while i < n:
        if is_img[i]:
            if g >= len(grids):
                raise ValueError("more image-token runs than images")
            t, h, w = grids[g]
            g += 1
            lh, lw = h // cfg.merge, w // cfg.merge
            cnt = t * lh * lw
            if i + cnt > n or not is_img[i:i + cnt].all():
                raise ValueError("image-token run does not match the image grid")
            tt = np.repeat(np.arange(t), lh * lw)
            hh = np.tile(np.repeat(np.arange(lh), lw), t)
            ww = np.tile(np.arange(lw), t * lh)
            pos.append(np.stack([tt, hh, ww]) + cur)
            cur += max(h, w) // cfg.merge
            i += cnt
        else:
            j = i
            while j < n and not is_img[j]:
                j += 1
            pos.append(np.broadcast_to(np.arange(j - i)[None, :], (3, j - i)) + cur)
            cur += j - i
            i = j
    if g != len(grids):
        raise ValueError("fewer image-token runs than images")
    p = np.concatenate(pos, axis=1).astype(np.int64)
    return p, int(p.max()) + 1


def mrope_cos_sin(cfg: Qwen2VLConfig, pos3: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """pos3 [3, S] -> cos/sin rows [S, head_dim] f32 with the (t,h,w) sections already selected per
    channel (TF modeling_qwen2_vl.py:156-170 + the cos.split(mrope_section*2) selection of :212-217)."""
    D = cfg.head_dim
    inv_freq = (1.0 / (np.float32(cfg.rope_theta) ** (np.arange(0, D, 2, dtype=np.float32) / np.float32(D))))
    inv_freq = inv_freq.astype(np.float32)
    axis = np.concatenate([np.full(s, i % 3) for i, s in enumerate(cfg.mrope_section)])  # [D/2]
    sel = pos3.astype(np.float32)[axis, :].T                                             # [S, D/2]
    freqs = sel * inv_freq[None, :]
    emb = np.concatenate([freqs, freqs], axis=-1)
    return np.cos(emb).astype(np.float32), np.sin(emb).astype(np.float32)


# the last chained decode call enqueued per device (Qwen2VLEngine.decode orders such calls on the GPU); one lock per device:
# engines on different GPUs driven from different threads do not serialise their host launch loops
_CHAIN_LOCKS_GUARD = threading.Lock()
_CHAIN_ORDER_LOCKS: Dict[Optional[int], threading.Lock] = {}
_CHAIN_LAST: Dict[Optional[int], "torch.cuda.Event"] = {}
_LOG = logging.getLogger("vision_inspection_system_amd.engine")


def _chain_order_lock(index: Optional[int]) -> threading.Lock:
    with _CHAIN_LOCKS_GUARD:
        return _CHAIN_ORDER_LOCKS.setdefault(index, threading.Lock())


# ----------------------------------------------------------------------------- engine
class Qwen2VLEngine:
    """One model replica on one GPU.  Not re-entrant: callers serialise through ``self.lock``."""

    def __init__(self, cfg: Qwen2VLConfig, weights: DeviceWeights, device, max_ctx: int = 4096,
                 decode_splits: int = 0, max_batch: int = 1, decode_weights: str = "bf16",
                 prefill_dtype: str = "bf16"):
        """decode_weights="fp8" (BASELINE configs[4] slice, also VIS_DECODE_WEIGHTS=fp8): the single-sequence decode
        step streams OCP-e4m3 copies of the LLM projections and the lm_head (per-output-row f32 scales,
        hip.quantize_fp8_rows at load time) through vis_gemv_fp8w; prefill and the batched decode keep bf16."""
        cfg.validate_for_kernels()
        hip.load()  # fail loudly when the gfx950 library is missing: there is no other path
        if not torch.cuda.is_available():
            raise hip.HipLibraryError("Qwen2VLEngine needs a ROCm GPU (no CPU fallback exists)")
        self.cfg, self.w, self.device = cfg, weights, torch.device(device)
        self.max_ctx = _round_up(max_ctx, 64)
        self.nsplit = decode_splits or max(1, -(-self.max_ctx // hip.DECODE_KEYS_PER_SPLIT))
        self.lock = threading.Lock()
        dev, bf = self.device, torch.bfloat16
        L, Hkv, Hq, D, H = cfg.layers, cfg.kv_heads, cfg.heads, cfg.head_dim, cfg.hidden
        if not 1 <= max_batch <= 64:
            raise ValueError("max_batch must be in 1..64 (one, two or four 16-row MFMA blocks of in-flight sequences)")
        Bm = self.max_batch = max_batch
        # per-sequence ("slot") state; slot 0 doubles as the single-sequence engine
        self.kcache_b = torch.zeros((Bm, L, Hkv, self.max_ctx, D), dtype=bf, device=dev)
        self.vcache_b = torch.zeros((Bm, L, Hkv, self.max_ctx, D), dtype=bf, device=dev)
        self.cos_b = torch.zeros((Bm, self.max_ctx, D), dtype=torch.float32, device=dev)
        self.sin_b = torch.zeros((Bm, self.max_ctx, D), dtype=torch.float32, device=dev)
        # decode-step state (device resident so the step is one replayable graph)
        self.step_b = torch.zeros(Bm, dtype=torch.int32, device=dev)
        self.cur_b = torch.zeros(Bm, dtype=torch.int32, device=dev)
        self.tokens_b = torch.zeros((Bm, self.max_ctx), dtype=torch.int32, device=dev)
        self.ws_val = torch.empty(max(256 * Bm, 2048), dtype=torch.float32, device=dev)
        self.ws_idx = torch.empty(max(256 * Bm, 2048), dtype=torch.int32, device=dev)
        self.logits_b = torch.empty((Bm, cfg.vocab), dtype=torch.float32, device=dev)
        self.kcache, self.vcache = self.kcache_b[0], self.vcache_b[0]
        self.cos_t, self.sin_t = self.cos_b[0], self.sin_b[0]
        self.step, self.cur_token = self.step_b[0:1], self.cur_b[0:1]
        self.tokens, self.logits = self.tokens_b[0], self.logits_b[0]
        nq = (Hq + 2 * Hkv) * D
        self.d_x = torch.empty((1, H), dtype=bf, device=dev)
        self.d_x2 = torch.empty((1, H), dtype=bf, device=dev)
        self.d_qkv = torch.empty(nq, dtype=bf, device=dev)
        self.d_attn = torch.empty(Hq * D, dtype=bf, device=dev)
        self.d_act = torch.empty(cfg.intermediate, dtype=bf, device=dev)
        self.part_o = torch.empty(Bm * Hq * self.nsplit * D, dtype=torch.float32, device=dev)
        self.part_ml = torch.empty(Bm * Hq * self.nsplit * 2, dtype=torch.float32, device=dev)
        if Bm > 1:  # batched-decode activations and the split-K partial workspace
            self.b_x = torch.empty((Bm, H), dtype=bf, device=dev)
            self.b_x2 = torch.empty((Bm, H), dtype=bf, device=dev)
            self.b_qkv = torch.empty((Bm, nq), dtype=bf, device=dev)
            self.b_attn = torch.empty((Bm, Hq * D), dtype=bf, device=dev)
            self.b_act = torch.empty((Bm, cfg.intermediate), dtype=bf, device=dev)
            self.b_xn = torch.empty((Bm, H), dtype=bf, device=dev)
            self.b_xn2 = torch.empty((Bm, H), dtype=bf, device=dev)
        # Batched decode, r05 experiment (VIS_DECODE_FUSED=1, OFF by default): every projection as ONE launch - stream + split-K
        # reduction + epilogue (vis_decode_proj_*: stream-K ranges with a last-arriver reduction, or whole-K column slabs), the
        # RMSNorm split into the producer's per-column and the consumer's per-row factor; csrc/decode_stream.hip,
        # csrc/decode_colpar.hip.  Correct and tested, but NOT faster: a cross-CU reduction costs three dependent memory round
        # trips wherever it runs, and the finalisation launch it replaces already sits at the ~5 us floor of a dependent
        # load -> store kernel (profiles/r05_decode_step_*.txt: 64 sequences 5.37 -> 6.4 ms per step).
        self.fused_proj = Bm > 1 and H % 128 == 0 and os.environ.get("VIS_DECODE_FUSED", "0") == "1"
        # batched decode: the self-attention launch finalises the qkv projection's partial slabs itself (vis_decode_attn_parts,
        # bit-identical to skinny_finalize + decode_attn); VIS_QKV_FOLD=0 keeps the two launches (A/B)
        self.fold_qkv = os.environ.get("VIS_QKV_FOLD", "1") == "1"
        # stacked suffix pass: rope / KV write / attention of the group's requests as ONE launch each (vis_*_many, per request
        # bit-identical to the per-request launches); VIS_GROUP_ATTN=0 keeps one launch per request (A/B)
        self.group_attn = os.environ.get("VIS_GROUP_ATTN", "1") == "1"
        if self.fused_proj:
            self.b_xw = torch.empty((Bm, H), dtype=bf, device=dev)        # x * ln1_w of the next layer (A operand of qkv / lm_head)
            self.b_x2w = torch.empty((Bm, H), dtype=bf, device=dev)       # x2 * ln2_w (A operand of gate/up)
            self.b_ssq1 = torch.zeros((H // hip.SSQ_UNIT, hip.SSQ_LD), dtype=torch.float32, device=dev)
            self.b_ssq2 = torch.zeros((H // hip.SSQ_UNIT, hip.SSQ_LD), dtype=torch.float32, device=dev)
        if Bm > 1 and not (self.fused_proj and decode_weights == "fp8"):
            # split-K slabs of the r02-r04 pair of launches: the whole step with VIS_DECODE_FUSED=0, and the bf16 down
            # projection at many sequences in the fused step (_decode_step_fused)
            self.b_part = torch.empty(16 * hip.part_rows(Bm) * max(nq, H, 2 * cfg.intermediate), dtype=torch.float32,
                                      device=dev)   # 16 stream-K slots x (16 or 32) rows
        self.decode_weights = decode_weights
        if decode_weights not in ("bf16", "fp8"):
            raise ValueError("decode_weights must be 'bf16' or 'fp8'")
        # Single-sequence decode: the head of every layer (qkv projection -> rope / append / attention -> o projection) as ONE
        # launch whose stages hand over inside the grid (csrc/decode_chain.hip; bit-identical to the four launches it
        # replaces).  VIS_DECODE_CHAIN=0 keeps the four launches (A/B).
        self.chain_sync: Optional[torch.Tensor] = None
        self._chain_state: Optional[tuple] = None      # (ws, sync) kept while the chain is switched off after a stall
        self._chain_launches = 0                       # host-side count of chained launches on the sync block (epoch guard)
        self._chain_clean_requests = 0                 # requests served on the four launches since a stall
        if decode_weights == "bf16" and os.environ.get("VIS_DECODE_CHAIN", "1") != "0" \
                and hip.decode_chain_supported(Hq, Hkv, D, H):
            self.chain_ws, self.chain_sync = hip.decode_chain_state(dev, Hq, Hkv, self.nsplit)
        # contexts up to this many cached keys decode on the chained launch, longer ones on the four launches (same bits):
        # what must be resident together are the workgroups that wait - projection and merge roles + Hkv attention items per
        # 64 keys of context (7B shapes: 6208 keys, whatever VIS_MAX_CTX is)
        self.chain_ctx_limit = hip.decode_chain_ctx_limit(Hq, Hkv, H) if self.chain_sync is not None else 0
        if self.chain_sync is not None and self.chain_ctx_limit <= 0:
            self.chain_sync = None
        self.q8: List[dict] = []
        self.prefill_dtype = prefill_dtype
        if prefill_dtype not in ("bf16", "fp8"):
            raise ValueError("prefill_dtype must be 'bf16' or 'fp8'")
        if decode_weights == "fp8" or prefill_dtype == "fp8":
            for lw in weights.llm:
                self.q8.append({n: hip.quantize_fp8_rows(getattr(lw, n)) for n in ("qkv_w", "o_w", "gateup_w", "down_w")})
            self.q8_lm_head = hip.quantize_fp8_rows(weights.lm_head)
        self.fp8_batched = False
        if decode_weights == "fp8" and Bm > 1 and cfg.hidden % 128 == 0:
            self.fp8_batched = True
            kp = _round_up(cfg.intermediate, 128)
            self.b_xq = torch.zeros((Bm, cfg.hidden), dtype=torch.uint8, device=dev)
            self.b_x2q = torch.zeros((Bm, cfg.hidden), dtype=torch.uint8, device=dev)
            self.b_actq = torch.zeros((Bm, kp), dtype=torch.uint8, device=dev)       # pad columns stay 0
            self.b_sx = torch.zeros((3, Bm), dtype=torch.float32, device=dev)
            if self.fused_proj:   # MX blocks: one E8M0 scale byte per row and 32 columns (pad blocks stay 0 = 2^-127 x code 0)
                self.b_xqs = torch.zeros((Bm, cfg.hidden // 32), dtype=torch.uint8, device=dev)
                self.b_x2qs = torch.zeros((Bm, cfg.hidden // 32), dtype=torch.uint8, device=dev)
                self.b_actqs = torch.zeros((Bm, kp // 32), dtype=torch.uint8, device=dev)
        if prefill_dtype == "fp8" or self.fp8_batched:   # fp8 MFMA needs K % 128 == 0: zero-pad down's K if necessary
            self.kpad = _round_up(cfg.intermediate, 128)
            if self.kpad != cfg.intermediate:
                for q in self.q8:
                    wq, sc = q["down_w"]
                    pad = torch.zeros((wq.shape[0], self.kpad), dtype=torch.uint8, device=wq.device)
                    pad[:, :cfg.intermediate] = wq
                    q["down_w_pad"] = (pad, sc)
            if prefill_dtype == "fp8" and (cfg.hidden % 128 or (cfg.heads * cfg.head_dim) % 128):
                raise ValueError("fp8 prefill needs hidden and heads*head_dim to be multiples of 128")
        if self.fused_proj:
            fp8b = self.fp8_batched
            kd = self.kpad if fp8b else cfg.intermediate
            shapes = [(nq, H, fp8b), (H, Hq * D, False), (2 * cfg.intermediate, H, fp8b), (H, kd, fp8b), (cfg.vocab, H, fp8b)]
            lib = hip.load()
            need = max(int(lib.vis_decode_proj_ws_bytes(Bm, n, k, 1 if f8 else 0)) for n, k, f8 in shapes)
            if need <= 0:
                raise ValueError("vis_decode_proj_ws_bytes refused a projection shape of this model")
            self.b_proj_ws = torch.zeros(need, dtype=torch.uint8, device=dev)
        self.vq8: List[dict] = []
        if prefill_dtype == "fp8" and cfg.v_mlp % 128 == 0 and os.environ.get("VIS_VIT_FP8", "1") == "1":
            # ViT block projections in e4m3 too (VIS_VIT_FP8=0 keeps the tower in bf16).  With only the 256x256 fp8 tile
            # this was slower than bf16 (M = 4900, N = 1280..5120 leaves 100-400 tiles for 256 CUs: 41.0 vs 40.3 ms);
            # with the 128x128 fp8 tile it pays: 39.8 -> 38.3 ms.  K = v_embed is zero-padded to a multiple of 128
            # when necessary (tiny: 320).
            self.vepad = _round_up(cfg.v_embed, 128)

            def q8pad(wt):
                wq, sc = hip.quantize_fp8_rows(wt)
                if wq.shape[1] % 128:
                    pad = torch.zeros((wq.shape[0], _round_up(wq.shape[1], 128)), dtype=torch.uint8, device=wq.device)
                    pad[:, :wq.shape[1]] = wq
                    wq = pad
                return wq, sc
            for b in weights.vit:
                self.vq8.append({n: q8pad(getattr(b, n)) for n in ("qkv_w", "proj_w", "fc1_w", "fc2_w")})
        self.slot_prompt_len = [0] * Bm
        # keys [0, batch_shared_len) of the CURRENT batch's slots are identical copies of one text prefix (prefill_many):
        # the batched decode attention reads them from slot 0 (VIS_DECODE_SHARED=0: every sequence reads its own copy)
        self.batch_shared_len = 0
        self._prefill_streams: List[torch.cuda.Stream] = []
        self.last_timing: dict = {}
        self._graphs: Dict[tuple, torch.cuda.CUDAGraph] = {}
        self.min_shared_prefix = 256     # shorter common prefixes are not worth a separate pass
        # K / V / V^T of text prefixes seen before (prompt caching across requests): the reference's agents put the same
        # ~1000-token inspection prompt in front of every image (vlm_inspector.py:452-470), so a later request - single or
        # batched - computes only its rows from the image on.  LRU; VIS_PREFIX_CACHE = entries kept (0 = off), ~82 MB each
        # at 7B shapes and 960 tokens.
        self._prefix_cache: "collections.OrderedDict[bytes, dict]" = collections.OrderedDict()
        self.prefix_cache_hits = 0
        self.temperature, self.seed = 0.0, 0
        self._vis_rope_cache: Dict[tuple, Tuple[torch.Tensor, torch.Tensor]] = {}
        self._rope_cache: Dict[tuple, tuple] = {}
        self._pairs_cache: Dict[tuple, torch.Tensor] = {}
        self._vis_work_cache: Dict[tuple, torch.Tensor] = {}
        self.prompt_len = 0
        self._decoded = 0
        self.decode_limit = 0
        self.last_first_logits: Optional[torch.Tensor] = None

    # ------------------------------------------------------------------ vision tower
    def vision_forward(self, frames: Sequence[torch.Tensor], split_rows: bool = True) -> torch.Tensor:
        """frames: uint8 device tensors [H, W, 3] (H, W multiples of 28) -> [n_image_tokens, hidden] bf16."""
        cfg, w, dev, bf = self.cfg, self.w, self.device, torch.bfloat16
        grids = [(1, f.shape[0] // cfg.patch, f.shape[1] // cfg.patch) for f in frames]
        counts = [g[1] * g[2] for g in grids]
        # Every image starts on a 64-row boundary: the attention kernel walks keys in absolute 64-row tiles, so an
        # image's online-softmax grouping (and with it the bf16 rounding of its features) would otherwise depend on
        # what precedes it in the batch.  With aligned starts an image's features are bit-identical whether it runs
        # alone, second in a request, or batched with other requests' images (tests/test_engine_gpu.py).  The pad
        # rows (<= 60 per image) are zero patches that nothing attends to; they are dropped from the result.
        starts, s0 = [], 0
        for c in counts:
            starts.append(s0)
            s0 += _round_up(c, 64)
        N = starts[-1] + counts[-1]
        padded = N != sum(counts)
        E, Hh, D = cfg.v_embed, cfg.v_heads, cfg.v_head_dim
        kp = w.patch_w.shape[1]
        patches = (torch.zeros if padded else torch.empty)((N, kp), dtype=bf, device=dev)
        for f, r0 in zip(frames, starts):
            hip.patchify(f, patches, r0, cfg.image_mean, cfg.image_std)
        x = hip.gemm(patches, w.patch_w)
        key = tuple(grids)
        if key not in self._vis_rope_cache:
            c, s = vision_cos_sin(cfg, grids)
            if padded:
                cp, sp = np.zeros((N, c.shape[1]), c.dtype), np.zeros((N, s.shape[1]), s.dtype)
                o = 0
                for r0, n in zip(starts, counts):
                    cp[r0:r0 + n], sp[r0:r0 + n] = c[o:o + n], s[o:o + n]
                    o += n
                c, s = cp, sp
            self._vis_rope_cache[key] = (torch.from_numpy(c).to(dev), torch.from_numpy(s).to(dev))
            if len(self._vis_rope_cache) > 16:
                self._vis_rope_cache.pop(next(iter(self._vis_rope_cache)))
        cos, sin = self._vis_rope_cache[key]
        if cfg.vision_arch == "qwen2_5_vl":
            return self._vision_forward_windowed(x, grids, starts, counts, N, padded, cos, sin)
        segs = [(r0, r0 + c) for r0, c in zip(starts, counts)]
        work = self._vis_work_cache.get(tuple(segs))
        if work is None:
            if len(self._vis_work_cache) >= 16:
                self._vis_work_cache.clear()
            # key-split plan for long segments (one 1024 x 1024 image: 9 of 39 row blocks); VIS_ATTN_SPLIT=0: plain items (A/B)
            work = self._vis_work_cache[tuple(segs)] = hip.make_vit_attn_plan(
                segs, dev, Hh, split=(D == 80 and os.environ.get("VIS_ATTN_SPLIT", "1") != "0"))
        ld = _round_up(N, 64)
        y = torch.empty((N, E), dtype=bf, device=dev)
        qkv = torch.empty((N, 3 * E), dtype=bf, device=dev)
        q = torch.empty((Hh, N, D), dtype=bf, device=dev)
        k = torch.empty((Hh, N, D), dtype=bf, device=dev)
        vt = torch.empty((Hh, D, ld), dtype=bf, device=dev)
        att = (torch.zeros if padded else torch.empty)((N, E), dtype=bf, device=dev)     # pad rows are never written
        hmid = torch.empty((N, cfg.v_mlp), dtype=bf, device=dev)
        scale = D ** -0.5
        merger_ln_done = False
        if self.vq8:
            # fp8 configuration: the four block projections on the fp8 MFMA, LayerNorm fused into the activation quantiser
            xq = torch.zeros((N, self.vepad), dtype=torch.uint8, device=dev)      # pad columns stay 0
            hq = torch.empty((N, cfg.v_mlp), dtype=torch.uint8, device=dev)
            sx = torch.empty(N, dtype=torch.float32, device=dev)
            for b, q8 in zip(w.vit, self.vq8):
                hip.quant_rows_fp8(x, xq, sx, norm_w=b.ln1_w, norm_b=b.ln1_b, eps=1e-6)
                hip.gemm_fp8(xq, sx, *q8["qkv_w"], bias=b.qkv_b, out=qkv)
                hip.qkv_rope_split(qkv, cos, sin, q, k, None, vt, Hh, Hh, D)
                hip.attn_prefill_plan(q, k, vt, att, work, scale)
                hip.quant_rows_fp8(att, xq, sx)
                hip.gemm_fp8(xq, sx, *q8["proj_w"], bias=b.proj_b, residual=x, out=x)
                hip.quant_rows_fp8(x, xq, sx, norm_w=b.ln2_w, norm_b=b.ln2_b, eps=1e-6)
                hip.gemm_fp8(xq, sx, *q8["fc1_w"], bias=b.fc1_b, act=hip.ACT_QUICKGELU, out=hmid)
                hip.quant_rows_fp8(hmid, hq, sx)
                hip.gemm_fp8(hq, sx, *q8["fc2_w"], bias=b.fc2_b, residual=x, out=x)
        else:
            # fc2 (K = 5120, only 100 tiles of 256 x 256 at M = 4900) runs as two K-slices (200 workgroups) whose
            # finalisation - sum + bias + residual - is fused with the NEXT LayerNorm (the next block's norm1, or the
            # merger's ln_q after the last block): the row owner exists there, so the norm costs no pass of its own.
            # The choice depends on the layer shape only, never on the row count (an image's features must not depend
            # on what shares the batch).  VIS_VIT_FC2_SPLITK=0: plain fc2 GEMM + separate LayerNorm (A/B).
            splitk = cfg.v_mlp >= 1024 and os.environ.get("VIS_VIT_FC2_SPLITK", "1") != "0"
            swork = torch.empty(2 * N * E, dtype=torch.float32, device=dev) if splitk else None
            nb = len(w.vit)
            hip.layernorm(x, w.vit[0].ln1_w, w.vit[0].ln1_b, 1e-6, out=y)
            for bi, b in enumerate(w.vit):
                hip.gemm(y, b.qkv_w, bias=b.qkv_b, out=qkv)
                hip.qkv_rope_split(qkv, cos, sin, q, k, None, vt, Hh, Hh, D)
                hip.attn_prefill_plan(q, k, vt, att, work, scale)
                hip.gemm(att, b.proj_w, bias=b.proj_b, residual=x, out=x)
                hip.layernorm(x, b.ln2_w, b.ln2_b, 1e-6, out=y)
                hip.gemm(y, b.fc1_w, bias=b.fc1_b, act=hip.ACT_QUICKGELU, out=hmid)
                nw, nbias = (w.vit[bi + 1].ln1_w, w.vit[bi + 1].ln1_b) if bi + 1 < nb else (w.merger_ln_w, w.merger_ln_b)
                if splitk:
                    hip.gemm_splitk_part(hmid, b.fc2_w, swork, 2)
                    hip.splitk_finalize_norm(swork, 2, x, bias=b.fc2_b, residual=x, norm_w=nw, norm_b=nbias, y_out=y,
                                             eps=1e-6)
                else:
                    hip.gemm(hmid, b.fc2_w, bias=b.fc2_b, residual=x, out=x)
                    hip.layernorm(x, nw, nbias, 1e-6, out=y)
            merger_ln_done = True
        if not merger_ln_done:
            hip.layernorm(x, w.merger_ln_w, w.merger_ln_b, 1e-6, out=y)
        m = cfg.merge ** 2
        z = hip.gemm(y.view(N // m, E * m), w.merger_fc0_w, bias=w.merger_fc0_b, act=hip.ACT_GELU_ERF)
        out = hip.gemm(z, w.merger_fc2_w, bias=w.merger_fc2_b)
        if padded:
            out = torch.cat([out[r0 // m:(r0 + c) // m] for r0, c in zip(starts, counts)])
        return out

    def _vision_forward_windowed(self, x, grids, starts, counts, N: int, padded: bool, cos, sin) -> torch.Tensor:
        """Qwen2.5-VL tower (TF:models/qwen2_5_vl/modeling_qwen2_5_vl.py:408-472) on the patch-embedded rows ``x``: the
        2x2 merge units of every image are permuted into window order (one row gather; the rope rows move with them),
        blocks = RMSNorm -> qkv -> 2-D rope -> attention over the WINDOWS (work items of <= 64 rows) or, in the
        v_fullatt blocks, over the whole image -> proj(+res) -> RMSNorm -> gate/up GEMM with bias + SwiGLU epilogue ->
        down projection as two K-slices finalised together with the next RMSNorm; merger (RMSNorm, GELU MLP); the merged
        rows are gathered back into the original order.  Same kernels as the Qwen2-VL tower."""
        cfg, w, dev, bf = self.cfg, self.w, self.device, torch.bfloat16
        E, Hh, D = cfg.v_embed, cfg.v_heads, cfg.v_head_dim
        m = cfg.merge ** 2
        key = ("win", tuple(grids))
        lay = self._vis_work_cache.get(key)
        if lay is None:
            row_idx = np.arange(N, dtype=np.int32)                 # pad rows (if any) stay where they are
            rev = np.arange(N // m, dtype=np.int32)
            win_segs, full_segs = [], []
            for g, r0, c in zip(grids, starts, counts):
                widx, cu = vision_window_order(cfg, g)
                rows = (widx[:, None] * m + np.arange(m)[None, :]).reshape(-1)
                row_idx[r0:r0 + c] = r0 + rows
                rev[r0 // m:(r0 + c) // m] = r0 // m + np.argsort(widx)
                win_segs += [(r0 + a, r0 + b) for a, b in zip(cu[:-1], cu[1:])]
                full_segs.append((r0, r0 + c))
            c_np, s_np = vision_cos_sin(cfg, grids)                 # rope rows follow their patches: permuted on the host
            cp, sp = np.zeros((N, c_np.shape[1]), np.float32), np.zeros((N, s_np.shape[1]), np.float32)
            o = 0
            for r0, c in zip(starts, counts):
                cp[r0:r0 + c], sp[r0:r0 + c] = c_np[o:o + c], s_np[o:o + c]
                o += c
            if len(self._vis_work_cache) >= 16:
                self._vis_work_cache.clear()
            lay = self._vis_work_cache[key] = (
                torch.from_numpy(row_idx).to(dev), torch.from_numpy(rev).to(dev),
                hip.make_vit_attn_plan(win_segs, dev, Hh, split=False),
                hip.make_vit_attn_plan(full_segs, dev, Hh, split=(D == 80 and os.environ.get("VIS_ATTN_SPLIT", "1") != "0")),
                torch.from_numpy(cp[row_idx]).to(dev), torch.from_numpy(sp[row_idx]).to(dev))
        row_idx, rev, work_win, work_full, cw, sw = lay
        xw = torch.empty_like(x)
        hip.gather_rows(x, row_idx, xw)
        x = xw
        ld = _round_up(N, 64)
        pad = cfg.v_mlp_pad
        y = torch.empty((N, E), dtype=bf, device=dev)
        qkv = torch.empty((N, 3 * E), dtype=bf, device=dev)
        q = torch.empty((Hh, N, D), dtype=bf, device=dev)
        k = torch.empty((Hh, N, D), dtype=bf, device=dev)
        vt = torch.empty((Hh, D, ld), dtype=bf, device=dev)
        att = (torch.zeros if padded else torch.empty)((N, E), dtype=bf, device=dev)
        hmid = torch.empty((N, pad), dtype=bf, device=dev)
        swork = torch.empty(2 * N * E, dtype=torch.float32, device=dev)
        scale = D ** -0.5
        nb = len(w.vit)
        hip.rmsnorm(x, w.vit[0].ln1_w, 1e-6, out=y)
        for bi, b in enumerate(w.vit):
            hip.gemm(y, b.qkv_w, bias=b.qkv_b, out=qkv)
            hip.qkv_rope_split(qkv, cw, sw, q, k, None, vt, Hh, Hh, D)
            hip.attn_prefill_plan(q, k, vt, att, work_full if bi in cfg.v_fullatt else work_win, scale)
            hip.gemm(att, b.proj_w, bias=b.proj_b, residual=x, out=x)
            hip.rmsnorm(x, b.ln2_w, 1e-6, out=y)
            hip.gemm(y, b.fc1_w, bias=b.fc1_b, act=hip.ACT_SWIGLU, out=hmid)
            nw = w.vit[bi + 1].ln1_w if bi + 1 < nb else w.merger_ln_w
            hip.gemm_splitk_part(hmid, b.fc2_w, swork, 2)
            hip.splitk_finalize_norm(swork, 2, x, bias=b.fc2_b, residual=x, norm_w=nw, y_out=y, eps=1e-6)
        z = hip.gemm(y.view(N // m, E * m), w.merger_fc0_w, bias=w.merger_fc0_b, act=hip.ACT_GELU_ERF)
        out_w = hip.gemm(z, w.merger_fc2_w, bias=w.merger_fc2_b)
        out = torch.empty_like(out_w)
        hip.gather_rows(out_w, rev, out)
        if padded:
            out = torch.cat([out[r0 // m:(r0 + c) // m] for r0, c in zip(starts, counts)])
        return out

    # ------------------------------------------------------------------ prefill
    def prefill(self, input_ids: Sequence[int], frames: Sequence[torch.Tensor] = (),
                ids_dev: Optional[torch.Tensor] = None, taps: Optional[dict] = None,
                temperature: float = 0.0, seed: int = 0, max_new_tokens: Optional[int] = None,
                slot: int = 0, split_vit: bool = True, image_embeds: Optional[torch.Tensor] = None,
                prefix: Optional[dict] = None, collect_prefix: bool = False) -> Optional[dict]:
        """Run the prompt through the LLM, fill the KV cache of ``slot`` and pick the first token
        (greedy when temperature == 0, Gumbel-max sampled otherwise).

        ``prefix`` (from ``collect_prefix``): K / V / V^T of the first P tokens (P a multiple of 64, text only) computed by
        another pass over the SAME leading tokens - the reference puts the ~1000-token inspection prompt in front of the
        image (src/agents/vlm_inspector.py:462-470), so the images of a batch share it.  They are copied into this slot
        and only rows P.. are computed; every row's arithmetic is unchanged (row-independent GEMMs, attention key tiles
        on absolute 64-key boundaries), so the result is bit-identical to the full pass.
        ``collect_prefix``: run rows 0..S-1 only (no lm_head, no token) and return that bundle."""
        cfg, w, dev, bf = self.cfg, self.w, self.device, torch.bfloat16
        if not 0 <= slot < self.max_batch:
            raise ValueError("slot out of range")
        self.temperature, self.seed = float(temperature), int(seed)
        kcache, vcache = self.kcache_b[slot], self.vcache_b[slot]
        cos_t, sin_t = self.cos_b[slot], self.sin_b[slot]
        step, cur_token = self.step_b[slot:slot + 1], self.cur_b[slot:slot + 1]
        tokens, logits = self.tokens_b[slot], self.logits_b[slot]
        S = len(input_ids)
        if S < 1 or S + 1 > self.max_ctx:
            raise ValueError(f"prompt of {S} tokens does not fit the context of {self.max_ctx}")
        ids_np = np.asarray(list(input_ids), dtype=np.int64)
        if ids_np.min() < 0 or ids_np.max() >= cfg.vocab:
            raise ValueError("token id out of range")
        grids = [(1, f.shape[0] // cfg.patch, f.shape[1] // cfg.patch) for f in frames]
        # decode rows: slot S + t carries rope position next_pos + t on all three axes
        n_dec = self.max_ctx - S if max_new_tokens is None else min(self.max_ctx - S, max_new_tokens + 1)
        # The M-RoPE tables (and the image-token scatter index) depend only on WHERE the image tokens sit, not on the
        # text: the images of a batch inspection - and every request with the same prompt template and frame size -
        # share them, so they are built once (host trigonometry + H2D) and kept on the device.
        is_img = ids_np == cfg.image_token_id
        key = (S, n_dec, tuple(grids), bool(is_img[0]), np.flatnonzero(np.diff(is_img.view(np.int8))).tobytes())
        hit = self._rope_cache.get(key)
        if hit is None:
            pos3, next_pos = rope_index(cfg, ids_np, grids)
            cos_np, sin_np = mrope_cos_sin(cfg, pos3)
            dpos = np.broadcast_to((next_pos + np.arange(n_dec))[None, :], (3, n_dec))
            dcos, dsin = mrope_cos_sin(cfg, dpos)
            img_idx = np.nonzero(is_img)[0].astype(np.int32)
            hit = (torch.from_numpy(np.concatenate([cos_np, dcos])).to(dev), torch.from_numpy(np.concatenate([sin_np, dsin])).to(dev),
                   torch.from_numpy(img_idx).to(dev))
            if len(self._rope_cache) >= 16:
                self._rope_cache.pop(next(iter(self._rope_cache)))
            self._rope_cache[key] = hit
        cos_t[:S + n_dec].copy_(hit[0], non_blocking=True)
        sin_t[:S + n_dec].copy_(hit[1], non_blocking=True)
        img_idx_dev = hit[2]
        if slot == 0:
            self.decode_limit = S + n_dec
        if ids_dev is None:
            ids_dev = hip.upload(ids_np.astype(np.int32), dev)
        H, Hq, Hkv, D = cfg.hidden, cfg.heads, cfg.kv_heads, cfg.head_dim
        L = len(w.llm)
        P = 0
        if prefix is not None:
            P = int(prefix["len"])
            if P % 64 or not 0 < P < S or collect_prefix:
                raise ValueError("bad shared prefix")
            if not np.array_equal(ids_np[:P], prefix["ids"]):
                raise ValueError("the shared prefix does not match this prompt")
            if (ids_np[:P] == cfg.image_token_id).any():
                raise ValueError("a shared prefix must be text only")
        n = S - P                                   # rows computed here
        x = torch.empty((n, H), dtype=bf, device=dev)
        hip.gather_rows(w.embed, ids_dev[P:] if P else ids_dev, x)
        if len(frames):
            # image_embeds: the merged ViT output of these frames computed elsewhere (prefill_many batches the tower
            # over several requests' images)
            img = image_embeds if image_embeds is not None else self.vision_forward(frames, split_rows=split_vit)
            if img_idx_dev.shape[0] != img.shape[0]:
                raise ValueError(f"image tokens ({img_idx_dev.shape[0]}) and image features ({img.shape[0]}) do not match")
            hip.scatter_rows(img, (img_idx_dev - P) if P else img_idx_dev, x)
            if taps is not None:
                taps["image_embeds"] = img
        cos, sin = cos_t[P:S], sin_t[P:S]
        # causal pass over rows P..S-1: 128-row query blocks paired latest-with-earliest, one pair per workgroup
        # (hip.attn_prefill_pairs); VIS_ATTN_PAIRS=0 keeps one block per workgroup (A/B: same results, bit for bit)
        pairs = os.environ.get("VIS_ATTN_PAIRS", "1") != "0"
        if pairs:
            work = self._pairs_cache.get((P, S))
            if work is None:
                if len(self._pairs_cache) >= 64:
                    self._pairs_cache.clear()
                work = self._pairs_cache[(P, S)] = hip.make_attn_pairs(P, S, dev)
        elif P:
            items = [(q0, min(128, S - q0), 0, S) for q0 in range(P, S, 128)]
            items.sort(key=lambda it: -(it[0] + it[1]))
            work = torch.tensor(items, dtype=torch.int32, device=dev).reshape(-1, 4).contiguous()
        else:
            work = hip.make_attn_work([(0, S)], True, dev)
        self._causal_pairs = pairs
        ld = _round_up(S, 64)
        nq = (Hq + 2 * Hkv) * D
        y = torch.empty((n, H), dtype=bf, device=dev)
        qkv = torch.empty((n, nq), dtype=bf, device=dev)
        q = torch.empty((Hq, n, D), dtype=bf, device=dev)
        # V^T per layer: one buffer re-used by all layers, or - when a prefix is involved - one per layer, so that the
        # prefix columns can be copied in (or handed out) in one go
        per_layer_vt = bool(P) or collect_prefix
        vt_all = torch.empty((L if per_layer_vt else 1, Hkv, D, ld), dtype=bf, device=dev)
        if P:
            kcache[:, :, :P].copy_(prefix["k"])
            vcache[:, :, :P].copy_(prefix["v"])
            vt_all[:, :, :, :P].copy_(prefix["vt"])
        att = torch.empty((n, Hq * D), dtype=bf, device=dev)
        act = torch.empty((n, cfg.intermediate), dtype=bf, device=dev)
        scale = D ** -0.5
        # The long-K down projection always runs as two K-slices (f32 partials, fixed-order sum): at S = 2249 its 126
        # 256x256 tiles would leave half the chip idle, and making the choice a function of the LAYER only (never of the
        # row count) keeps every row's summation order the same in the full, the prefix and the suffix pass.
        splitk_work = None
        if cfg.intermediate >= 8192 and H % 8 == 0:
            splitk_work = torch.empty(2 * n * H, dtype=torch.float32, device=dev)
        if self.prefill_dtype == "fp8":
            self._llm_layers_fp8(x, qkv, q, vt_all, att, act, cos, sin, kcache, vcache, work, n, taps, P)
        else:
            hip.rmsnorm(x, w.llm[0].ln1_w, cfg.rms_eps, out=y)
            for li, lw in enumerate(w.llm):
                vt = vt_all[li if per_layer_vt else 0]
                hip.gemm(y, lw.qkv_w, bias=lw.qkv_b, out=qkv)
                hip.qkv_rope_split(qkv, cos, sin, q, kcache[li], vcache[li], vt, Hq, Hkv, D, k_pos0=P, vt_col0=P)
                if pairs:
                    hip.attn_prefill_pairs(q, kcache[li], vt, att, work, scale, q_row0=P)
                else:
                    hip.attn_prefill(q, kcache[li], vt, att, work, True, scale, q_row0=P)
                hip.gemm(att, lw.o_w, residual=x, out=x)
                hip.rmsnorm(x, lw.ln2_w, cfg.rms_eps, out=y)
                hip.gemm(y, lw.gateup_w, act=hip.ACT_SWIGLU, out=act)
                nxt = w.llm[li + 1].ln1_w if li + 1 < L else None      # the next layer's input norm rides on the finalisation
                if splitk_work is not None:      # long K, too few 256x256 tiles for the chip: two K-slices per tile
                    hip.gemm_splitk_part(act, lw.down_w, splitk_work, 2)
                    hip.splitk_finalize_norm(splitk_work, 2, x, residual=x, norm_w=nxt, y_out=y if nxt is not None else None,
                                             eps=cfg.rms_eps)
                else:
                    hip.gemm(act, lw.down_w, residual=x, out=x)
                    if nxt is not None:
                        hip.rmsnorm(x, nxt, cfg.rms_eps, out=y)
                if taps is not None and li == 0:
                    taps["layer0"] = x.clone()
        if collect_prefix:
            Pn = (S // 64) * 64
            return {"len": Pn, "ids": ids_np[:Pn].copy(), "k": kcache[:, :, :Pn].clone(), "v": vcache[:, :, :Pn].clone(),
                    "vt": vt_all[:, :, :, :Pn].clone()}
        # first token: final norm fused into the lm_head GEMV of the last position only
        hip.gemv(x[n - 1], w.lm_head, logits, norm_w=w.final_norm_w, eps=cfg.rms_eps)
        if taps is not None:
            taps["first_logits"] = logits.clone()
        step.fill_(S - 1)
        hip.argmax(logits, self.ws_val[256 * slot:256 * (slot + 1)], self.ws_idx[256 * slot:256 * (slot + 1)], tokens,
                   cur_token, step, self.temperature, self.seed + 0x9E3779B9 * slot)   # per-slot workspace: prefills
        # of different slots may run concurrently on different streams
        self.slot_prompt_len[slot] = S
        if slot == 0:
            self.prompt_len = S
            self._decoded = 0
        return None

    def _llm_layers_fp8(self, x, qkv, q, vt_all, att, act, cos, sin, kcache, vcache, work, S, taps, P: int = 0) -> None:
        """The decoder layers of the prompt pass with every projection on the fp8 MFMA (configs[4]).  S = rows computed
        here, P = rows of a shared prefix already in the cache (vt_all: one V^T buffer per layer when it has > 1)."""
        cfg, w, dev = self.cfg, self.w, self.device
        H, Hq, Hkv, D = cfg.hidden, cfg.heads, cfg.kv_heads, cfg.head_dim
        scale = D ** -0.5
        xq = torch.empty((S, H), dtype=torch.uint8, device=dev)
        aq = torch.empty((S, Hq * D), dtype=torch.uint8, device=dev)
        hq = torch.zeros((S, self.kpad), dtype=torch.uint8, device=dev)      # pad columns stay 0 (= +0.0 in e4m3)
        sx = torch.empty(S, dtype=torch.float32, device=dev)
        dwork = torch.empty(2 * S * H, dtype=torch.float32, device=dev) \
            if (self.kpad >= 8192 and H % 8 == 0) else None      # as in the bf16 pass: a function of the layer only
        for li, lw in enumerate(w.llm):
            q8 = self.q8[li]
            vt = vt_all[li if vt_all.shape[0] > 1 else 0]
            hip.quant_rows_fp8(x, xq, sx, norm_w=lw.ln1_w, eps=cfg.rms_eps)
            hip.gemm_fp8(xq, sx, *q8["qkv_w"], bias=lw.qkv_b, out=qkv)
            hip.qkv_rope_split(qkv, cos, sin, q, kcache[li], vcache[li], vt, Hq, Hkv, D, k_pos0=P, vt_col0=P)
            if self._causal_pairs:
                hip.attn_prefill_pairs(q, kcache[li], vt, att, work, scale, q_row0=P)
            else:
                hip.attn_prefill(q, kcache[li], vt, att, work, True, scale, q_row0=P)
            hip.quant_rows_fp8(att, aq, sx)
            hip.gemm_fp8(aq, sx, *q8["o_w"], residual=x, out=x)
            hip.quant_rows_fp8(x, xq, sx, norm_w=lw.ln2_w, eps=cfg.rms_eps)
            hip.gemm_fp8(xq, sx, *q8["gateup_w"], act=hip.ACT_SWIGLU, out=act)
            hip.quant_rows_fp8(act, hq[:, :cfg.intermediate], sx)
            hip.gemm_fp8(hq, sx, *q8.get("down_w_pad", q8["down_w"]), residual=x, out=x, work=dwork, ksplit=2)
            if taps is not None and li == 0:
                taps["layer0"] = x.clone()

    def _prefill_group(self, items: Sequence[tuple], prefix: dict, temperature: float, seed: int,
                       max_new_tokens: Optional[int]) -> None:
        """The prompt pass of SEVERAL requests that share one text prefix and one prompt structure (the images of a batch
        inspection: same template, same frame size) as ONE pass over their stacked suffix rows.

        items: [(slot, input_ids, ids_dev or None, image_embeds)] - all prompts have the same length S and the image tokens
        in the same places.  Per request n = S - P rows (7B bench prompt: 1289 = 5.04 row tiles of 256: a sixth of the LLM
        GEMM tiles is padding, and the qkv / o projections fill 0.42 / 0.33 of a round of the chip); stacked, k requests
        give k n rows (four: 20.1 tiles, 1.5 / 1.15 rounds).  Only the row-independent kernels see the stack - embedding
        gather, projections, norms, finalisations; rope / KV write / attention run per request on its own rows, KV-cache
        slot and V^T buffer.  Every row's arithmetic is what the single-request pass does (a projection's K order does
        not depend on M, the down projection is two K-slices either way, key tiles are absolute), so tokens and logits
        are bit-identical to it (tests/test_engine_gpu.py, test_fullsize_gpu.py)."""
        cfg, w, dev, bf = self.cfg, self.w, self.device, torch.bfloat16
        self.temperature, self.seed = float(temperature), int(seed)
        k = len(items)
        S = len(items[0][1])
        P = int(prefix["len"])
        n = S - P
        H, Hq, Hkv, D = cfg.hidden, cfg.heads, cfg.kv_heads, cfg.head_dim
        L = len(w.llm)
        ids0 = np.asarray(list(items[0][1]), dtype=np.int64)
        is_img = ids0 == cfg.image_token_id
        grids_n = int(is_img.sum())
        n_dec = self.max_ctx - S if max_new_tokens is None else min(self.max_ctx - S, max_new_tokens + 1)
        ld = _round_up(S, 64)
        nq = (Hq + 2 * Hkv) * D
        x = torch.empty((k * n, H), dtype=bf, device=dev)
        tabs = None
        for j, (slot, ids, ids_dev, img) in enumerate(items):
            ids_np = np.asarray(list(ids), dtype=np.int64)
            if len(ids_np) != S or not np.array_equal(ids_np == cfg.image_token_id, is_img) or \
                    not np.array_equal(ids_np[:P], prefix["ids"]) or img.shape[0] != grids_n:
                raise ValueError("_prefill_group: the prompts of a group must share length, image positions and prefix")
            if ids_np.min() < 0 or ids_np.max() >= cfg.vocab:
                raise ValueError("token id out of range")
            if tabs is None:     # rope tables / scatter index: one structure for the whole group (cached per structure)
                key = ("grp", S, n_dec, grids_n, bool(is_img[0]), np.flatnonzero(np.diff(is_img.view(np.int8))).tobytes())
                tabs = self._rope_cache.get(key)
                if tabs is None:
                    raise ValueError("_prefill_group: rope tables missing (prefill_many prepares them)")
            self.cos_b[slot][:S + n_dec].copy_(tabs[0], non_blocking=True)
            self.sin_b[slot][:S + n_dec].copy_(tabs[1], non_blocking=True)
            if ids_dev is None:
                ids_dev = hip.upload(ids_np.astype(np.int32), dev)
            xj = x[j * n:(j + 1) * n]
            hip.gather_rows(w.embed, ids_dev[P:], xj)
            hip.scatter_rows(img, tabs[2] - P, xj)
            self.kcache_b[slot][:, :, :P].copy_(prefix["k"])
            self.vcache_b[slot][:, :, :P].copy_(prefix["v"])
        s0 = items[0][0]                      # every slot of the group now holds the same (contiguous) tables
        cos, sin = self.cos_b[s0][P:S], self.sin_b[s0][P:S]
        work = self._pairs_cache.get((P, S))
        if work is None:
            if len(self._pairs_cache) >= 64:
                self._pairs_cache.clear()
            work = self._pairs_cache[(P, S)] = hip.make_attn_pairs(P, S, dev)
        y = torch.empty((k * n, H), dtype=bf, device=dev)
        qkv = torch.empty((k * n, nq), dtype=bf, device=dev)
        q = torch.empty((k, Hq, n, D), dtype=bf, device=dev)
        vt_all = torch.empty((k, L, Hkv, D, ld), dtype=bf, device=dev)
        for j in range(k):
            vt_all[j][:, :, :, :P].copy_(prefix["vt"])
        att = torch.empty((k * n, Hq * D), dtype=bf, device=dev)
        act = torch.empty((k * n, cfg.intermediate), dtype=bf, device=dev)
        scale = D ** -0.5
        splitk_work = torch.empty(2 * k * n * H, dtype=torch.float32, device=dev) \
            if (cfg.intermediate >= 8192 and H % 8 == 0 and self.prefill_dtype != "fp8") else None

        slots = [it[0] for it in items]
        T = self.kcache_b.shape[3]
        many = self.group_attn and k > 1 and k <= hip.MAX_GROUP_REQUESTS and self.kcache_b.is_contiguous()

        def attend(li):        # rope / KV write / attention: per request, on its rows, cache slot and V^T buffer
            if many:           # ... as ONE launch each for the whole group (r05: a 1289-row suffix alone fills a third of the chip)
                kv_off = [sl * self.kcache_b.stride(0) + li * self.kcache_b.stride(1) for sl in slots]
                vt_l = vt_all[:, li]
                hip.qkv_rope_split_many(qkv, cos, sin, q, self.kcache_b, self.vcache_b, vt_l, Hq, Hkv, D, kv_off, T,
                                        k_pos0=P, vt_col0=P)
                hip.attn_prefill_pairs_many(q, self.kcache_b, vt_l, att, work, scale, kv_off, T, q_row0=P)
                return
            for j, (slot, _, _, _) in enumerate(items):
                rows = slice(j * n, (j + 1) * n)
                kc, vc, vt = self.kcache_b[slot][li], self.vcache_b[slot][li], vt_all[j][li]
                hip.qkv_rope_split(qkv[rows], cos, sin, q[j], kc, vc, vt, Hq, Hkv, D, k_pos0=P, vt_col0=P)
                hip.attn_prefill_pairs(q[j], kc, vt, att[rows], work, scale, q_row0=P)

        if self.prefill_dtype == "fp8":
            # configs[4]: every projection on the fp8 MFMA; the per-token activation quantiser is row-wise, so it stacks
            # like the projections do (the layer body of _llm_layers_fp8 over k n rows)
            M = k * n
            xq = torch.empty((M, H), dtype=torch.uint8, device=dev)
            aq = torch.empty((M, Hq * D), dtype=torch.uint8, device=dev)
            hq = torch.zeros((M, self.kpad), dtype=torch.uint8, device=dev)
            sx = torch.empty(M, dtype=torch.float32, device=dev)
            dwork = torch.empty(2 * M * H, dtype=torch.float32, device=dev) if (self.kpad >= 8192 and H % 8 == 0) else None
            for li, lw in enumerate(w.llm):
                q8 = self.q8[li]
                hip.quant_rows_fp8(x, xq, sx, norm_w=lw.ln1_w, eps=cfg.rms_eps)
                hip.gemm_fp8(xq, sx, *q8["qkv_w"], bias=lw.qkv_b, out=qkv)
                attend(li)
                hip.quant_rows_fp8(att, aq, sx)
                hip.gemm_fp8(aq, sx, *q8["o_w"], residual=x, out=x)
                hip.quant_rows_fp8(x, xq, sx, norm_w=lw.ln2_w, eps=cfg.rms_eps)
                hip.gemm_fp8(xq, sx, *q8["gateup_w"], act=hip.ACT_SWIGLU, out=act)
                hip.quant_rows_fp8(act, hq[:, :cfg.intermediate], sx)
                hip.gemm_fp8(hq, sx, *q8.get("down_w_pad", q8["down_w"]), residual=x, out=x, work=dwork, ksplit=2)
        else:
            hip.rmsnorm(x, w.llm[0].ln1_w, cfg.rms_eps, out=y)
        for li, lw in enumerate(w.llm if self.prefill_dtype != "fp8" else ()):
            hip.gemm(y, lw.qkv_w, bias=lw.qkv_b, out=qkv)
            attend(li)
            hip.gemm(att, lw.o_w, residual=x, out=x)
            hip.rmsnorm(x, lw.ln2_w, cfg.rms_eps, out=y)
            hip.gemm(y, lw.gateup_w, act=hip.ACT_SWIGLU, out=act)
            nxt = w.llm[li + 1].ln1_w if li + 1 < L else None
            if splitk_work is not None:
                hip.gemm_splitk_part(act, lw.down_w, splitk_work, 2)
                hip.splitk_finalize_norm(splitk_work, 2, x, residual=x, norm_w=nxt, y_out=y if nxt is not None else None,
                                         eps=cfg.rms_eps)
            else:
                hip.gemm(act, lw.down_w, residual=x, out=x)
                if nxt is not None:
                    hip.rmsnorm(x, nxt, cfg.rms_eps, out=y)
        # lm_head over the last rows of the whole group in ONE pass over its 1.09 GB (vis_gemv_bf16_rows: every row bit-identical to
        # the single-row GEMV) instead of one pass per request; VIS_GROUP_ATTN=0 keeps the per-request calls
        grouped = self.group_attn and 1 < k <= 4 and H * 2 * (2 if k <= 2 else 4) <= 152 * 1024
        if grouped:
            lg = torch.empty((k, self.logits_b.shape[1]), dtype=torch.float32, device=dev)
            hip.gemv_rows(x[n - 1::n], w.lm_head, lg, norm_w=w.final_norm_w, eps=cfg.rms_eps)
        for j, (slot, _, _, _) in enumerate(items):
            logits = self.logits_b[slot]
            if grouped:
                logits.copy_(lg[j])
            else:
                hip.gemv(x[(j + 1) * n - 1], w.lm_head, logits, norm_w=w.final_norm_w, eps=cfg.rms_eps)
            self.step_b[slot:slot + 1].fill_(S - 1)
            hip.argmax(logits, self.ws_val[256 * slot:256 * (slot + 1)], self.ws_idx[256 * slot:256 * (slot + 1)],
                       self.tokens_b[slot], self.cur_b[slot:slot + 1], self.step_b[slot:slot + 1], self.temperature,
                       self.seed + 0x9E3779B9 * slot)
            self.slot_prompt_len[slot] = S
            if slot == 0:
                self.prompt_len = S
                self._decoded = 0
                self.decode_limit = S + n_dec

    def _group_tables(self, ids: Sequence[int], n_img_tokens: int, max_new_tokens: Optional[int], frames) -> None:
        """Rope tables + image-token scatter index of a prompt structure, under the key _prefill_group looks up."""
        cfg, dev = self.cfg, self.device
        ids_np = np.asarray(list(ids), dtype=np.int64)
        S = len(ids_np)
        is_img = ids_np == cfg.image_token_id
        n_dec = self.max_ctx - S if max_new_tokens is None else min(self.max_ctx - S, max_new_tokens + 1)
        key = ("grp", S, n_dec, n_img_tokens, bool(is_img[0]), np.flatnonzero(np.diff(is_img.view(np.int8))).tobytes())
        if key in self._rope_cache:
            return
        grids = [(1, f.shape[0] // cfg.patch, f.shape[1] // cfg.patch) for f in frames]
        pos3, next_pos = rope_index(cfg, ids_np, grids)
        cos_np, sin_np = mrope_cos_sin(cfg, pos3)
        dpos = np.broadcast_to((next_pos + np.arange(n_dec))[None, :], (3, n_dec))
        dcos, dsin = mrope_cos_sin(cfg, dpos)
        img_idx = np.nonzero(is_img)[0].astype(np.int32)
        if len(self._rope_cache) >= 16:
            self._rope_cache.pop(next(iter(self._rope_cache)))
        self._rope_cache[key] = (torch.from_numpy(np.concatenate([cos_np, dcos])).to(dev),
                                 torch.from_numpy(np.concatenate([sin_np, dsin])).to(dev), torch.from_numpy(img_idx).to(dev))

    def prefill_many(self, requests: Sequence, temperature: float = 0.0,
                     seed: int = 0, max_new_tokens: Optional[int] = None,
                     ids_dev: Optional[Sequence[torch.Tensor]] = None) -> Tuple[List[Optional[int]], List[Optional[Exception]]]:
        """Prefill the requests into consecutive slots.  The prefills are independent kernel chains: they are issued round-robin
        on a few HIP streams (VIS_PREFILL_STREAMS, default 2: 418 -> 381 ms for 8 images) so that the ragged last round of one image's GEMM /
        attention grids is filled by another image's workgroups.  Returns with the current stream ordered after
        all of them.

        A request is ``(input_ids, frames)`` or a zero-argument callable returning that pair (LAZY form, used by the
        batch seam: the callable waits for the image's host decode and uploads it, so the GPU starts on image 0 while
        images 1.. are still being decoded).  Lazy requests are resolved in order, a ViT group at a time; one that
        raises gets no slot and its exception is returned instead of failing the batch.
        Returns (slot of request b or None, exception of request b or None)."""
        B = len(requests)
        lazy = any(callable(r) for r in requests)
        resolved: List[Optional[tuple]] = [None] * B
        errors: List[Optional[Exception]] = [None] * B
        slots: List[Optional[int]] = [None] * B

        def get(b):
            if resolved[b] is None and errors[b] is None:
                r = requests[b]
                if callable(r):
                    try:
                        r = r()
                    except Exception as e:      # noqa: BLE001 - stays this request's own failure
                        errors[b] = e
                        return None
                resolved[b] = r
            return resolved[b]

        n_streams = max(1, min(B, int(os.environ.get("VIS_PREFILL_STREAMS", "2"))))
        vb = max(1, int(os.environ.get("VIS_VIT_BATCH", "4"))) if n_streams > 1 else 1
        # The text in front of the first image is the same for every image of a batch inspection (the reference's
        # INSPECTOR_PROMPT, vlm_inspector.py:452-470): its K / V / V^T are computed once and copied into every slot.
        # Eager requests: the prefix common to ALL prompts; lazy requests: the prefix common to the first group (the
        # others are not known yet) - a later prompt that does not start with it is simply computed in full.
        first = [r for r in (get(b) for b in range(B if not lazy else min(B, max(vb, 2)))) if r is not None]
        shared = None
        P = self.shared_prefix_len([r[0] for r in first]) if (B > 1 and len(first) > 1) else 0
        if P:
            shared = self.cached_prefix(first[0][0], P, temperature, seed)
        unshared = [False]        # set when some slot of this batch did NOT take the shared prefix

        def prefix_for(ids):
            if shared is None or len(ids) <= P:
                unshared[0] = True
                return None
            hit = np.array_equal(np.asarray(list(ids[:P]), dtype=np.int64), shared["ids"])
            if not hit:
                unshared[0] = True
            return shared if hit else None

        self.batch_shared_len = 0

        next_slot = 0
        if n_streams == 1:
            for b in range(B):
                r = get(b)
                if r is None:
                    continue
                self.prefill(r[0], r[1], ids_dev=ids_dev[b] if ids_dev else None, temperature=temperature, seed=seed,
                             max_new_tokens=max_new_tokens, slot=next_slot, prefix=prefix_for(r[0]))
                slots[b] = next_slot
                next_slot += 1
            if shared is not None and not unshared[0] and os.environ.get("VIS_DECODE_SHARED", "1") != "0":
                self.batch_shared_len = P
            return slots, errors
        cur = torch.cuda.current_stream(self.device)
        if len(self._prefill_streams) < n_streams:
            self._prefill_streams = [torch.cuda.Stream(device=self.device) for _ in range(n_streams)]
        streams = self._prefill_streams[:n_streams]
        cfg = self.cfg
        for g0 in range(0, B, vb):
            grp_all = [b for b in range(g0, min(B, g0 + vb)) if get(b) is not None]
            # One ViT pass over the images of up to VIS_VIT_BATCH requests (default 4): at M = 4 x 4900 rows the tower's
            # GEMM grids become whole rounds of the chip (fc1 6.02, fc2 3.0, proj 3.0, qkv 9.02 rounds instead of
            # 1.56 / 0.78 / 0.76 / 2.29) and the varlen attention kernel takes the images as segments of one launch.
            embeds = {}
            grp = [b for b in grp_all if len(resolved[b][1])]
            if vb > 1 and len(grp) >= 2:
                out = self.vision_forward([f for b in grp for f in resolved[b][1]], split_rows=False)
                r0 = 0
                for b in grp:
                    n = sum((f.shape[0] // cfg.patch) * (f.shape[1] // cfg.patch) // cfg.merge ** 2 for f in resolved[b][1])
                    embeds[b] = out[r0:r0 + n]
                    r0 += n
            # Requests of the group that share the text prefix AND the prompt structure (the images of a batch inspection)
            # run their suffix rows as ONE stacked pass (_prefill_group: full LLM GEMM tiles and rounds instead of
            # 5.04-tile, 0.4-round grids per image); VIS_MERGE_PREFILL=0 keeps one pass per request (A/B, same results).
            merged: List[int] = []
            if shared is not None and os.environ.get("VIS_MERGE_PREFILL", "1") != "0":
                cand = [b for b in grp_all if b in embeds and prefix_for(resolved[b][0]) is not None]
                if len(cand) >= 2:
                    first_ids = np.asarray(list(resolved[cand[0]][0]), dtype=np.int64)
                    same = [b for b in cand if len(resolved[b][0]) == len(first_ids)
                            and embeds[b].shape[0] == embeds[cand[0]].shape[0]
                            and [f.shape for f in resolved[b][1]] == [f.shape for f in resolved[cand[0]][1]]
                            and np.array_equal(np.asarray(list(resolved[b][0]), dtype=np.int64) == cfg.image_token_id,
                                               first_ids == cfg.image_token_id)]
                    if len(same) >= 2:
                        merged = same
            slot_of = {b: next_slot + i for i, b in enumerate(grp_all)}       # slots follow the request order
            next_slot += len(grp_all)
            if merged:
                st = streams[(g0 // vb) % n_streams]         # consecutive groups alternate streams
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    items = []
                    for b in merged:
                        ids, frames = resolved[b]
                        for f in frames:
                            f.record_stream(st)
                        embeds[b].record_stream(st)
                        items.append((slot_of[b], ids, ids_dev[b] if ids_dev else None, embeds[b]))
                        slots[b] = slot_of[b]
                    self._group_tables(resolved[merged[0]][0], int(embeds[merged[0]].shape[0]), max_new_tokens,
                                       resolved[merged[0]][1])
                    self._prefill_group(items, shared, temperature, seed, max_new_tokens)
            for b in grp_all:
                if b in merged:
                    continue
                ids, frames = resolved[b]
                st = streams[slot_of[b] % n_streams]
                st.wait_stream(cur)              # the frames' upload / resize and the group's ViT pass ran on `cur`
                with torch.cuda.stream(st):
                    for f in frames:
                        f.record_stream(st)
                    if b in embeds:
                        embeds[b].record_stream(st)
                    self.prefill(ids, frames, ids_dev=ids_dev[b] if ids_dev else None, temperature=temperature, seed=seed,
                                 max_new_tokens=max_new_tokens, slot=slot_of[b], split_vit=False, image_embeds=embeds.get(b),
                                 prefix=prefix_for(ids))
                slots[b] = slot_of[b]
        for st in streams:
            cur.wait_stream(st)
        if shared is not None:
            for t in (shared["k"], shared["v"], shared["vt"]):
                for st in streams:
                    t.record_stream(st)
        if shared is not None and not unshared[0] and os.environ.get("VIS_DECODE_SHARED", "1") != "0":
            self.batch_shared_len = P
        return slots, errors

    def text_prefix_len(self, ids: Sequence[int]) -> int:
        """Length (a multiple of 64, 0 = none worth keeping) of the text-only prefix of ONE prompt: everything in front of its
        first image, leaving at least one row to compute."""
        cfg = self.cfg
        a = np.asarray(list(ids), dtype=np.int64)
        special = np.nonzero((a == cfg.image_token_id) | (a == cfg.vision_start_id))[0]
        P = int(special[0]) if special.size else len(a) - 1
        P = (min(P, len(a) - 1) // 64) * 64
        return P if P >= self.min_shared_prefix else 0

    def cached_prefix(self, ids: Sequence[int], P: int, temperature: float = 0.0, seed: int = 0) -> Optional[dict]:
        """The K / V / V^T bundle of ``ids[:P]`` (P from text_prefix_len / shared_prefix_len): from the cache, or computed
        now (one pass over those rows - no more work than the full pass would have spent on them) and kept."""
        keep = int(os.environ.get("VIS_PREFIX_CACHE", "4"))
        if P <= 0:
            return None
        key = np.asarray(list(ids[:P]), dtype=np.int64).tobytes()
        hit = self._prefix_cache.get(key) if keep > 0 else None
        if hit is not None:
            self._prefix_cache.move_to_end(key)
            self.prefix_cache_hits += 1
            return hit
        bundle = self.prefill(list(ids[:P]), (), temperature=temperature, seed=seed, max_new_tokens=0, slot=0,
                              collect_prefix=True)
        if keep > 0:
            self._prefix_cache[key] = bundle
            while len(self._prefix_cache) > keep:
                self._prefix_cache.popitem(last=False)
        return bundle

    def shared_prefix_len(self, id_lists: Sequence[Sequence[int]]) -> int:
        """Length (a multiple of 64, 0 = do not share) of the text-only token prefix common to all prompts of a batch."""
        if len(id_lists) < 2 or os.environ.get("VIS_SHARE_PREFIX", "1") == "0":
            return 0
        cfg = self.cfg
        first = np.asarray(list(id_lists[0]), dtype=np.int64)
        P = len(first) - 1                       # every request keeps at least one row of its own
        for ids in id_lists[1:]:
            other = np.asarray(list(ids), dtype=np.int64)
            m = min(P, len(other) - 1)
            neq = np.nonzero(first[:m] != other[:m])[0]
            P = int(neq[0]) if neq.size else m
        special = np.nonzero((first[:P] == cfg.image_token_id) | (first[:P] == cfg.vision_start_id))[0]
        if special.size:
            P = int(special[0])
        P = (P // 64) * 64
        return P if P >= self.min_shared_prefix else 0

    # ------------------------------------------------------------------ decode
    def _decode_step(self, chained: Optional[bool] = None) -> None:
        cfg, w = self.cfg, self.w
        Hq, Hkv, D = cfg.heads, cfg.kv_heads, cfg.head_dim
        scale = D ** -0.5
        if chained is None:
            chained = self.chain_sync is not None
        chained = chained and self.chain_sync is not None
        if not chained:
            hip.gather_rows(w.embed, self.cur_token, self.d_x)
        x, x2 = self.d_x, self.d_x2
        if self.decode_weights == "fp8":
            for li, lw in enumerate(w.llm):
                q = self.q8[li]
                hip.gemv_fp8(x[0], *q["qkv_w"], self.d_qkv, bias=lw.qkv_b, norm_w=lw.ln1_w, eps=cfg.rms_eps)
                hip.decode_attn(self.d_qkv, self.cos_t, self.sin_t, self.kcache[li], self.vcache[li], self.step,
                                self.part_o, self.part_ml, self.d_attn, Hq, Hkv, D, self.nsplit, scale)
                hip.gemv_fp8(self.d_attn, *q["o_w"], x2[0], residual=x[0])
                hip.gemv_fp8(x2[0], *q["gateup_w"], self.d_act, norm_w=lw.ln2_w, act=hip.ACT_SWIGLU, eps=cfg.rms_eps)
                hip.gemv_fp8(self.d_act, *q["down_w"], x[0], residual=x2[0])
            hip.gemv_fp8(x[0], *self.q8_lm_head, self.logits, norm_w=w.final_norm_w, eps=cfg.rms_eps)
            hip.argmax(self.logits, self.ws_val, self.ws_idx, self.tokens, self.cur_token, self.step,
                       self.temperature, self.seed)
            return
        for li, lw in enumerate(w.llm):
            if chained:
                try:   # layer 0 reads the new token's embedding row itself (x_index): no gather launch
                    hip.decode_chain(w.embed if li == 0 else x[0], lw.qkv_w, lw.qkv_b, lw.ln1_w, lw.o_w, x2[0], self.cos_t,
                                     self.sin_t, self.kcache[li], self.vcache[li], self.step, self.chain_ws, self.chain_sync,
                                     Hq, Hkv, D, self.nsplit, scale, cfg.rms_eps, x_index=self.cur_token if li == 0 else None,
                                     ctx_bound=self.chain_ctx_limit)
                except hip.ChainRefused as e:
                    # VIS_ERR_UNSUPPORTED and nothing else (a bad argument or a launch error propagates): the grid for this
                    # context length is larger than the device holds resident -> the four launches, same results; said once
                    if li:
                        raise
                    _LOG.warning("%s - decoding on the four launches per layer head (VIS_MAX_CTX=%d)", e, self.max_ctx)
                    self.chain_sync, self._chain_state, chained = None, None, False
                    hip.gather_rows(w.embed, self.cur_token, self.d_x)
            if not chained:
                hip.gemv(x[0], lw.qkv_w, self.d_qkv, bias=lw.qkv_b, norm_w=lw.ln1_w, eps=cfg.rms_eps)
                hip.decode_attn(self.d_qkv, self.cos_t, self.sin_t, self.kcache[li], self.vcache[li], self.step,
                                self.part_o, self.part_ml, self.d_attn, Hq, Hkv, D, self.nsplit, scale)
                hip.gemv(self.d_attn, lw.o_w, x2[0], residual=x[0])
            hip.gemv(x2[0], lw.gateup_w, self.d_act, norm_w=lw.ln2_w, act=hip.ACT_SWIGLU, eps=cfg.rms_eps)
            hip.gemv(self.d_act, lw.down_w, x[0], residual=x2[0])
        if chained:     # the pick's first stage rides in the lm_head epilogue
            hip.gemv_argmax(x[0], w.lm_head, self.logits, self.ws_val, self.ws_idx, self.tokens, self.cur_token, self.step,
                            norm_w=w.final_norm_w, eps=cfg.rms_eps, temperature=self.temperature, seed=self.seed)
            return
        hip.gemv(x[0], w.lm_head, self.logits, norm_w=w.final_norm_w, eps=cfg.rms_eps)
        hip.argmax(self.logits, self.ws_val, self.ws_idx, self.tokens, self.cur_token, self.step,
                   self.temperature, self.seed)

    # ---- batched decode: B in-flight sequences (slots 0..B-1) share every weight read of a step
    def _decode_step_batched(self, B: int) -> None:
        """Every projection = gemm_decode (weights streamed once for all B sequences, split-K f32 partials) +
        skinny_finalize (row-wise: sum, bias/residual/SwiGLU, and the RMSNorm of the NEXT projection)."""
        # VIS_ROWS_GEMV=<n> (default 0 = off): batches of 2..n (<= 4) sequences decode on the multi-row GEMV instead.  Measured
        # at exact 7B shapes (tools/probes/rows_ab.sh): 2 sequences 3.66 -> 3.20 ms per step (bf16), 2.83 -> 2.65 (e4m3
        # weights); at 3 and 4 the four-row kernel LOSES to the stream-K projection (4.2-4.3 vs 3.7-3.8 ms; fp8 3.8-4.0 vs
        # 2.9: four LDS reads + four dot products per weight chunk, one workgroup per CU on the down projection).  Off by
        # default because it trades an invariant for those 6-13 %: a sequence decoded in a batch of 2 would then follow the
        # single-sequence arithmetic and in a batch of 3+ the stream-K arithmetic - its tokens could depend on the batch size
        # at near-ties (tests/test_fullsize_gpu.py::test_7b_batch_invariance_and_reproducibility).
        rows_max = int(os.environ.get("VIS_ROWS_GEMV", "0"))
        if 2 <= B <= min(rows_max, 4) and max(self.cfg.intermediate, self.cfg.hidden) * 2 * (2 if B <= 2 else 4) <= 152 * 1024:
            return self._decode_step_rows(B)
        if self.fused_proj:
            return self._decode_step_fused(B)
        if self.decode_weights == "fp8" and self.fp8_batched:
            return self._decode_step_batched_fp8(B)
        cfg, w = self.cfg, self.w
        Hq, Hkv, D = cfg.heads, cfg.kv_heads, cfg.head_dim
        scale, eps = D ** -0.5, cfg.rms_eps
        x, x2, qkv, att, act = self.b_x[:B], self.b_x2[:B], self.b_qkv[:B], self.b_attn[:B], self.b_act[:B]
        xn, xn2, part = self.b_xn[:B], self.b_xn2[:B], self.b_part
        nq = qkv.shape[1]
        hip.gather_rows(w.embed, self.cur_b[:B], x)
        hip.rmsnorm(x, w.llm[0].ln1_w, eps, out=xn)
        n_layers = len(w.llm)
        for li, lw in enumerate(w.llm):
            ks = hip.decode_gemm(xn, lw.qkv_w, part=part)
            if self.fold_qkv:     # the attention workgroups finalise the qkv columns they read (same bits, one launch less)
                hip.decode_attn_parts(part, ks, self.cos_b[:B], self.sin_b[:B], self.kcache_b[:B, li], self.vcache_b[:B, li],
                                      self.step_b[:B], self.part_o, self.part_ml, att, Hq, Hkv, D, self.nsplit, scale,
                                      bias=lw.qkv_b, shared_len=self.batch_shared_len)
            else:
                hip.skinny_finalize(part, ks, qkv, nq, bias=lw.qkv_b, eps=eps)
                hip.decode_attn(qkv, self.cos_b[:B], self.sin_b[:B], self.kcache_b[:B, li], self.vcache_b[:B, li],
                                self.step_b[:B], self.part_o, self.part_ml, att, Hq, Hkv, D, self.nsplit, scale,
                                shared_len=self.batch_shared_len)
            ks = hip.decode_gemm(att, lw.o_w, part=part)
            hip.skinny_finalize(part, ks, x2, cfg.hidden, residual=x, norm_w=lw.ln2_w, yn=xn2, eps=eps)
            ks = hip.decode_gemm(xn2, lw.gateup_w, part=part)
            hip.skinny_finalize(part, ks, act, 2 * cfg.intermediate, swiglu=True, eps=eps)
            ks = hip.decode_gemm(act, lw.down_w, part=part)
            next_norm = w.llm[li + 1].ln1_w if li + 1 < n_layers else w.final_norm_w
            hip.skinny_finalize(part, ks, x, cfg.hidden, residual=x2, norm_w=next_norm, yn=xn, eps=eps)
        hip.decode_gemm(xn, w.lm_head, out=self.logits_b[:B])
        hip.argmax(self.logits_b[:B], self.ws_val, self.ws_idx, self.tokens_b[:B], self.cur_b[:B], self.step_b[:B],
                   self.temperature, self.seed)

    def _decode_step_fused(self, B: int, projections_only: bool = False) -> int:
        """One decode step for B in-flight sequences, every projection ONE launch (r05, csrc/decode_stream.hip): the stream-K
        weight pass, the fixed-order sum of a cut tile's segments by the last workgroup to arrive, and the epilogue - bias (qkv),
        SwiGLU (gate/up), residual + the NEXT norm's weight + the tile's sum of squares (o, down); the consumer of a normed row
        applies rs[b] = rsqrt(mean(x^2) + eps) to its finished sums (the RMSNorm, TF modeling_qwen2_vl.py:96-110, split into a
        per-column and a per-row factor).  5 launches per layer (r04: 9).  fp8 (configs[4]): activations travel as MX blocks
        written by the producing epilogue.  ``projections_only``: bench.py's replay of the weight-streaming launches alone.
        Returns the number of projection launches."""
        cfg, w = self.cfg, self.w
        Hq, Hkv, D, H = cfg.heads, cfg.kv_heads, cfg.head_dim, cfg.hidden
        scale, eps = D ** -0.5, cfg.rms_eps
        x, x2, qkv, att, act = self.b_x[:B], self.b_x2[:B], self.b_qkv[:B], self.b_attn[:B], self.b_act[:B]
        xw, x2w, s1, s2, ws = self.b_xw[:B], self.b_x2w[:B], self.b_ssq1, self.b_ssq2, self.b_proj_ws
        fp8 = self.decode_weights == "fp8" and self.fp8_batched
        if fp8:
            xq, xqs, x2q, x2qs = self.b_xq[:B], self.b_xqs[:B], self.b_x2q[:B], self.b_x2qs[:B]
            aq, aqs = self.b_actq[:B], self.b_actqs[:B]
        # Long-K, few-column projections at many sequences (7B down at 33+ sequences: K = 18944, 28 column tiles) keep the
        # r02-r04 pair - stream-K slabs + the row-owning finalisation, which then also applies the next RMSNorm itself (xw holds
        # the NORMALISED row, its consumer takes no row factor): a column slab per workgroup would stream 2.4 MB of x each, and
        # the in-kernel reduction of the stream-K form costs more than the finalisation launch it saves (64 vs 33 + 5 us,
        # profiles/r05_decode_step_b64_streamk.txt).  VIS_DOWN_PAIR=0 forces the single launch (A/B).
        down_pair = (not fp8) and hasattr(self, "b_part") and os.environ.get("VIS_DOWN_PAIR", "1") != "0" and \
            hip.decode_proj_form(B, H, cfg.intermediate, hip.DP_RESID_NORMW, False, False) == "streamk"
        s1_in = None if down_pair else s1
        if not projections_only:
            if down_pair:
                hip.gather_rows(w.embed, self.cur_b[:B], x)
                hip.rmsnorm(x, w.llm[0].ln1_w, eps, out=xw)
            else:
                hip.decode_prep_rows(w.embed, self.cur_b[:B], w.llm[0].ln1_w, x, None if fp8 else xw, s1,
                                     xq if fp8 else None, xqs if fp8 else None)
        n_layers = len(w.llm)
        for li, lw in enumerate(w.llm):
            next_norm = w.llm[li + 1].ln1_w if li + 1 < n_layers else w.final_norm_w
            if fp8:
                q8 = self.q8[li]
                hip.decode_proj_fp8(xq, xqs, *q8["qkv_w"], ws, hip.DP_PLAIN, out=qkv, bias=lw.qkv_b, ssq_in=s1, norm_dim=H, eps=eps)
            else:
                hip.decode_proj(xw, lw.qkv_w, ws, hip.DP_PLAIN, out=qkv, bias=lw.qkv_b, ssq_in=s1_in, norm_dim=H, eps=eps)
            if not projections_only:
                hip.decode_attn(qkv, self.cos_b[:B], self.sin_b[:B], self.kcache_b[:B, li], self.vcache_b[:B, li],
                                self.step_b[:B], self.part_o, self.part_ml, att, Hq, Hkv, D, self.nsplit, scale,
                                shared_len=self.batch_shared_len)
            if fp8:     # the o projection keeps bf16 weights (its input comes from the attention kernel); its epilogue writes MX
                hip.decode_proj(att, lw.o_w, ws, hip.DP_RESID_NORMW, out=x2, out_q=x2q, out_qs=x2qs, residual=x,
                                norm_w=lw.ln2_w, ssq_out=s2)
                hip.decode_proj_fp8(x2q, x2qs, *q8["gateup_w"], ws, hip.DP_SWIGLU, out_q=aq, out_qs=aqs, ssq_in=s2,
                                    norm_dim=H, eps=eps)
                hip.decode_proj_fp8(aq, aqs, *q8.get("down_w_pad", q8["down_w"]), ws, hip.DP_RESID_NORMW, out=x, out_q=xq,
                                    out_qs=xqs, residual=x2, norm_w=next_norm, ssq_out=s1)
            else:
                hip.decode_proj(att, lw.o_w, ws, hip.DP_RESID_NORMW, out=x2, out_w=x2w, residual=x, norm_w=lw.ln2_w, ssq_out=s2)
                hip.decode_proj(x2w, lw.gateup_w, ws, hip.DP_SWIGLU, out=act, ssq_in=s2, norm_dim=H, eps=eps)
                if down_pair:
                    ks = hip.decode_gemm(act, lw.down_w, part=self.b_part)
                    if not projections_only:
                        hip.skinny_finalize(self.b_part, ks, x, H, residual=x2, norm_w=next_norm, yn=xw, eps=eps)
                else:
                    hip.decode_proj(act, lw.down_w, ws, hip.DP_RESID_NORMW, out=x, out_w=xw, residual=x2, norm_w=next_norm,
                                    ssq_out=s1)
        if fp8:
            hip.decode_proj_fp8(xq, xqs, *self.q8_lm_head, ws, hip.DP_PLAIN, out=self.logits_b[:B], ssq_in=s1, norm_dim=H, eps=eps)
        else:
            hip.decode_proj(xw, w.lm_head, ws, hip.DP_PLAIN, out=self.logits_b[:B], ssq_in=s1_in, norm_dim=H, eps=eps)
        if not projections_only:
            hip.argmax(self.logits_b[:B], self.ws_val, self.ws_idx, self.tokens_b[:B], self.cur_b[:B], self.step_b[:B],
                       self.temperature, self.seed)
        return 4 * n_layers + 1

    def _decode_step_rows(self, B: int) -> None:
        """A couple of in-flight sequences: the single-sequence step with the multi-row GEMV (vis_gemv_*_rows) - one pass
        over the weights for all rows, norm / bias / residual / SwiGLU fused as at B = 1, no partial buffers and no
        finalisation launches (the stream-K projection pays four of those per layer); bf16 or e4m3 weights with bf16
        activations, every sequence bit-identical to decoding alone.  Chosen by _decode_step_batched (VIS_ROWS_GEMV)."""
        cfg, w = self.cfg, self.w
        Hq, Hkv, D = cfg.heads, cfg.kv_heads, cfg.head_dim
        scale, eps = D ** -0.5, cfg.rms_eps
        x, x2, qkv, att, act = self.b_x[:B], self.b_x2[:B], self.b_qkv[:B], self.b_attn[:B], self.b_act[:B]
        fp8 = self.decode_weights == "fp8"

        def proj(inp, wt, out, **kw):
            if fp8:
                hip.gemv_fp8_rows(inp, *wt, out, **kw)
            else:
                hip.gemv_rows(inp, wt, out, **kw)

        hip.gather_rows(w.embed, self.cur_b[:B], x)
        for li, lw in enumerate(w.llm):
            q8 = self.q8[li] if fp8 else None
            pick = (lambda n: q8[n]) if fp8 else (lambda n: getattr(lw, n))
            proj(x, pick("qkv_w"), qkv, bias=lw.qkv_b, norm_w=lw.ln1_w, eps=eps)
            hip.decode_attn(qkv, self.cos_b[:B], self.sin_b[:B], self.kcache_b[:B, li], self.vcache_b[:B, li],
                            self.step_b[:B], self.part_o, self.part_ml, att, Hq, Hkv, D, self.nsplit, scale,
                            shared_len=self.batch_shared_len)
            proj(att, pick("o_w"), x2, residual=x)
            proj(x2, pick("gateup_w"), act, norm_w=lw.ln2_w, act=hip.ACT_SWIGLU, eps=eps)
            proj(act, pick("down_w"), x, residual=x2)
        proj(x, self.q8_lm_head if fp8 else w.lm_head, self.logits_b[:B], norm_w=w.final_norm_w, eps=eps)
        hip.argmax(self.logits_b[:B], self.ws_val, self.ws_idx, self.tokens_b[:B], self.cur_b[:B], self.step_b[:B],
                   self.temperature, self.seed)

    def _decode_step_batched_fp8(self, B: int) -> None:
        """Batched decode on e4m3 weights AND activations (BASELINE configs[4]): qkv, gate/up, down and the lm_head run
        on the fp8 stream-K projection; every finalisation also emits the next projection's input as e4m3 + row scale
        (it owns the row, so the activation quantiser costs no extra launch).  The o projection stays bf16: its input
        comes from the attention combine, which does not own whole rows."""
        cfg, w = self.cfg, self.w
        Hq, Hkv, D = cfg.heads, cfg.kv_heads, cfg.head_dim
        scale, eps = D ** -0.5, cfg.rms_eps
        x, x2, qkv, att, act = self.b_x[:B], self.b_x2[:B], self.b_qkv[:B], self.b_attn[:B], self.b_act[:B]
        xn, xn2, part = self.b_xn[:B], self.b_xn2[:B], self.b_part
        xq, x2q, aq = self.b_xq[:B], self.b_x2q[:B], self.b_actq[:B]
        sxq, sx2q, saq = self.b_sx[0, :B], self.b_sx[1, :B], self.b_sx[2, :B]
        nq = qkv.shape[1]
        hip.gather_rows(w.embed, self.cur_b[:B], x)
        hip.quant_rows_fp8(x, xq, sxq, norm_w=w.llm[0].ln1_w, eps=eps)
        n_layers = len(w.llm)
        for li, lw in enumerate(w.llm):
            q8 = self.q8[li]
            ks = hip.decode_gemm_fp8(xq, sxq, *q8["qkv_w"], part=part)
            if self.fold_qkv:
                hip.decode_attn_parts(part, ks, self.cos_b[:B], self.sin_b[:B], self.kcache_b[:B, li], self.vcache_b[:B, li],
                                      self.step_b[:B], self.part_o, self.part_ml, att, Hq, Hkv, D, self.nsplit, scale,
                                      bias=lw.qkv_b, sx=sxq, sw=q8["qkv_w"][1], shared_len=self.batch_shared_len)
            else:
                hip.skinny_finalize_fp8(part, ks, qkv, nq, sx=sxq, sw=q8["qkv_w"][1], bias=lw.qkv_b, eps=eps)
                hip.decode_attn(qkv, self.cos_b[:B], self.sin_b[:B], self.kcache_b[:B, li], self.vcache_b[:B, li],
                                self.step_b[:B], self.part_o, self.part_ml, att, Hq, Hkv, D, self.nsplit, scale,
                                shared_len=self.batch_shared_len)
            ks = hip.decode_gemm(att, lw.o_w, part=part)
            hip.skinny_finalize_fp8(part, ks, x2, cfg.hidden, residual=x, norm_w=lw.ln2_w, yn=xn2, yq=x2q,
                                    yq_scale=sx2q, eps=eps)
            ks = hip.decode_gemm_fp8(x2q, sx2q, *q8["gateup_w"], part=part)
            hip.skinny_finalize_fp8(part, ks, act, 2 * cfg.intermediate, sx=sx2q, sw=q8["gateup_w"][1], swiglu=True,
                                    yq=aq, yq_scale=saq, eps=eps)
            dw = q8.get("down_w_pad", q8["down_w"])
            ks = hip.decode_gemm_fp8(aq, saq, *dw, part=part)
            next_norm = w.llm[li + 1].ln1_w if li + 1 < n_layers else w.final_norm_w
            hip.skinny_finalize_fp8(part, ks, x, cfg.hidden, sx=saq, sw=dw[1], residual=x2, norm_w=next_norm, yn=xn,
                                    yq=xq, yq_scale=sxq, eps=eps)
        hip.decode_gemm_fp8(xq, sxq, *self.q8_lm_head, out=self.logits_b[:B])
        hip.argmax(self.logits_b[:B], self.ws_val, self.ws_idx, self.tokens_b[:B], self.cur_b[:B], self.step_b[:B],
                   self.temperature, self.seed)

    def _ensure_graph(self, batch: int = 0, chained: bool = False) -> torch.cuda.CUDAGraph:
        # sampling parameters (and the batch size) are kernel arguments baked into the graph
        key = (self.temperature, self.seed, batch, self.batch_shared_len if batch else 0, bool(chained) and not batch)
        if key in self._graphs:
            return self._graphs[key]
        step_fn = (lambda: self._decode_step_batched(batch)) if batch else (lambda: self._decode_step(chained))
        # warm the kernels outside capture, then restore the counters the warm-up advanced
        saved = (self.step_b.clone(), self.cur_b.clone())
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step_fn()
        torch.cuda.current_stream().wait_stream(side)
        self.step_b.copy_(saved[0])
        self.cur_b.copy_(saved[1])
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):     # another agent's thread may be allocating
            step_fn()
        # capture does not execute; state is unchanged
        if len(self._graphs) >= 8:
            self._graphs.pop(next(iter(self._graphs)))
        self._graphs[key] = g
        return g

    def decode(self, n_steps: int, use_graph: bool = True) -> None:
        """Generate n_steps further tokens (each replays the captured step)."""
        if self.prompt_len + n_steps + 1 > self.max_ctx:
            raise ValueError("decode would overflow the KV cache")
        if self.prompt_len + self._decoded + n_steps > self.decode_limit:
            raise ValueError("decode beyond the rope rows prepared by prefill (pass max_new_tokens)")
        base = self.prompt_len + self._decoded       # step k of this call sees base + k cached keys
        self._decoded += n_steps
        n_chain = max(0, min(n_steps, self.chain_ctx_limit - base)) if self.chain_sync is not None else 0
        if n_chain == 0:
            return self._decode_steps(n_steps, use_graph, False)
        # Chained launches wait inside the grid for workgroups of the SAME launch, which only works while that launch can
        # have its whole grid resident (860 of the device's 1024 slots at 7B shapes).  Two engines decoding on two streams at
        # once could strand each other (each holding slots the other's producers need; the bounded waits would then raise).
        # So the chained decode calls of a device are ordered on the GPU: a call waits for the previous call's last launch
        # (an event, no host blocking), whatever streams or threads they come from.  Decode is HBM-bound: nothing is lost.
        self._chain_epoch_guard(n_chain)
        with _chain_order_lock(self.device.index):
            cur = torch.cuda.current_stream(self.device)
            prev = _CHAIN_LAST.get(self.device.index)
            if prev is not None:
                cur.wait_event(prev)
            self._decode_steps(n_chain, use_graph, True)
            ev = torch.cuda.Event()
            ev.record(cur)
            _CHAIN_LAST[self.device.index] = ev
        if n_steps > n_chain:       # the context has outgrown the chained grid: the four launches from here on (same bits)
            self._decode_steps(n_steps - n_chain, use_graph, False)

    def _chain_epoch_guard(self, n_steps: int) -> None:
        """The granule tag of a chained launch is the sync block's 32-bit launch counter + 1.  Long before it can wrap
        (2^31 launches = ~2.6 days of continuous single-sequence decode) workspace and sync block are zeroed on the stream,
        between two decode calls - no launch of this engine is in flight then (same stream), and the kernel itself skips
        tag 0, the mark of a never-written granule (tests/test_kernels_gpu.py::test_decode_chain_epoch_wrap)."""
        self._chain_launches += n_steps * len(self.w.llm)
        if self._chain_launches >= (1 << 31):
            self.chain_sync.zero_()
            self.chain_ws.zero_()
            self._chain_launches = n_steps * len(self.w.llm)

    def _decode_steps(self, n_steps: int, use_graph: bool, chained: bool = False) -> None:
        if use_graph:
            g = self._ensure_graph(0, chained)
            if chained and self.chain_sync is None:      # the warm-up met a refusal: the captured step is the unchained one
                g = self._ensure_graph(0, False)
            for _ in range(n_steps):
                g.replay()
        else:
            for _ in range(n_steps):
                self._decode_step(chained)

    def generated(self, n: int) -> List[int]:
        s = self.prompt_len - 1
        toks = self.tokens[s:s + n].cpu().tolist()
        self.check_chain()
        return toks

    def check_chain(self) -> None:
        """Raise if a bounded wait inside a chained layer-head launch gave up (its results are invalid); the sync block is
        zeroed so that the engine stays usable.  Called after the token D2H, i.e. when the stream has drained."""
        if self.chain_sync is not None and int(self.chain_sync[hip.CHAIN_STATUS_WORD].item()) != 0:
            self.chain_sync.zero_()
            self.chain_ws.zero_()
            self._chain_launches = 0
            raise hip.ChainStalled("vis_decode_chain: a hand-off wait inside the launch timed out (decode results invalid)")

    def disable_chain(self) -> None:
        """Back to the four launches per layer head (same results bit for bit); the captured decode graphs hold chained launches
        and are dropped.  The (zeroed) workspace is kept: `_maybe_reenable_chain` switches back after VIS_CHAIN_RETRY_AFTER
        requests served cleanly on the four launches - one transient stall (a co-tenant's burst) does not cost the engine
        its faster step for good."""
        if self.chain_sync is not None:
            self._chain_state = (self.chain_ws, self.chain_sync)
        self.chain_sync = None
        self._chain_clean_requests = 0
        self._graphs.clear()

    def _maybe_reenable_chain(self) -> None:
        if self.chain_sync is not None or self._chain_state is None:
            return
        self._chain_clean_requests += 1
        retry_after = int(os.environ.get("VIS_CHAIN_RETRY_AFTER", "64"))
        if retry_after > 0 and self._chain_clean_requests >= retry_after:
            self.chain_ws, self.chain_sync = self._chain_state
            self._chain_state = None
            self.chain_sync.zero_()
            self.chain_ws.zero_()
            self._chain_launches = 0
            self._graphs.clear()
            _LOG.warning("vis_decode_chain: %d requests served on the four launches since the stall - chained layer head back on",
                         self._chain_clean_requests)

    def generate(self, input_ids: Sequence[int], frames: Sequence[torch.Tensor] = (), max_new_tokens: int = 128,
                 ignore_eos: bool = False, use_graph: bool = True, check_every: int = 16,
                 temperature: float = 0.0, seed: int = 0) -> List[int]:
        """Generate up to max_new_tokens (greedy at temperature 0).  EOS is checked on the host every
        ``check_every`` tokens so the decode loop itself never synchronises; output is truncated at the
        first EOS (exclusive)."""
        room = self.max_ctx - len(input_ids) - 1
        if max_new_tokens > room and not getattr(self, "_warned_clamp", False):
            self._warned_clamp = True          # said once per engine: the reply may end before the model is done
            _LOG.warning("max_tokens=%d does not fit the context (prompt %d + reply <= VIS_MAX_CTX=%d): generating at most %d",
                         max_new_tokens, len(input_ids), self.max_ctx, max(1, room))
        max_new_tokens = max(1, min(max_new_tokens, room))
        try:
            out = self._generate(input_ids, frames, max_new_tokens, ignore_eos, use_graph, check_every, temperature, seed)
            self._maybe_reenable_chain()
            return out
        except hip.ChainStalled as e:
            # Something else held CU slots this launch's producers needed (another PROCESS sharing the GPU, or other work of
            # this process on another stream: chained launches of this process are ordered, Qwen2VLEngine.decode, everything
            # else is covered by the bounded wait only).  From the launch after the stall on every chained launch of the
            # request returned at once (status word read at kernel entry), so what was lost is one wait bound.  The request is
            # served again on the unchained launches - the same HIP kernels' arithmetic, identical tokens - and the engine
            # stays on them for the next VIS_CHAIN_RETRY_AFTER requests.
            _LOG.warning("%s - continuing on the unchained decode step", e)
            self.disable_chain()
            return self._generate(input_ids, frames, max_new_tokens, ignore_eos, use_graph, check_every, temperature, seed)

    def _generate(self, input_ids, frames, max_new_tokens, ignore_eos, use_graph, check_every, temperature, seed) -> List[int]:
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]      # per-stage device time (tracing)
        ev[0].record()
        # the text in front of the image (the agents' fixed inspection prompt): its K / V come from the prefix cache
        P = self.text_prefix_len(input_ids) if os.environ.get("VIS_SHARE_PREFIX", "1") != "0" else 0
        prefix = self.cached_prefix(input_ids, P, temperature, seed) if P else None
        self.prefill(input_ids, frames, temperature=temperature, seed=seed, max_new_tokens=max_new_tokens, prefix=prefix)
        ev[1].record()
        done, eos = 1, set(self.cfg.eos_ids)
        while done < max_new_tokens:
            if not ignore_eos:
                toks = self.generated(done)
                if any(t in eos for t in toks):
                    break
            n = min(check_every if not ignore_eos else max_new_tokens, max_new_tokens - done)
            self.decode(n, use_graph=use_graph)
            done += n
        ev[2].record()
        toks = self.generated(done)                                         # D2H: synchronises, the events have completed
        self.last_timing = {"prompt_tokens": len(input_ids), "prefill_ms": ev[0].elapsed_time(ev[1]),
                            "decode_ms": ev[1].elapsed_time(ev[2]), "decode_steps": done - 1, "sequences": 1}
        if not ignore_eos:
            for i, t in enumerate(toks):
                if t in eos:
                    return toks[:i]
        return toks

    # ------------------------------------------------------------------ batched generation
    def generate_batch(self, requests: Sequence,
                       max_new_tokens: int = 128, ignore_eos: bool = False, use_graph: bool = True,
                       check_every: int = 16, temperature: float = 0.0, seed: int = 0) -> list:
        """requests: [(input_ids, frames)] for up to max_batch images - or zero-argument callables returning that pair
        (see prefill_many: resolved in order while the GPU already works on the earlier ones).  Prefill runs per image
        (M = S rows is already MFMA-efficient); the decode steps are shared: one weight pass per step for all sequences.
        Returns one token list per request; for a lazy request whose callable raised, the exception object instead."""
        n_req = len(requests)
        if not 1 <= n_req <= self.max_batch:
            raise ValueError(f"batch of {n_req} does not fit max_batch={self.max_batch}")
        if n_req == 1:
            r = requests[0]
            if callable(r):
                try:
                    r = r()
                except Exception as e:      # noqa: BLE001
                    return [e]
            return [self.generate(r[0], r[1], max_new_tokens, ignore_eos, use_graph, check_every, temperature, seed)]
        # every prompt's own limit (prompt + new tokens <= context) is applied by its prefill; the shared loop below
        # runs to the limit of the longest one
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record()
        slots, errors = self.prefill_many(requests, temperature=temperature, seed=seed, max_new_tokens=max_new_tokens)
        ev[1].record()
        live = [b for b in range(n_req) if slots[b] is not None]
        B = len(live)
        if B == 0:
            return list(errors)
        longest = max(self.slot_prompt_len[slots[b]] for b in live)
        max_new_tokens = max(1, min(max_new_tokens, self.max_ctx - longest - 1))
        eos = set(self.cfg.eos_ids)
        starts = [self.slot_prompt_len[s] - 1 for s in range(B)]

        def collect(n):
            t = self.tokens_b[:B].cpu()
            return [t[b, starts[b]:starts[b] + n].tolist() for b in range(B)]

        done = 1
        g = self._ensure_graph(B) if use_graph else None
        while done < max_new_tokens:
            if not ignore_eos and all(any(t in eos for t in seq) for seq in collect(done)):
                break
            n = min(check_every if not ignore_eos else max_new_tokens, max_new_tokens - done)
            for _ in range(n):
                if g is not None:
                    g.replay()
                else:
                    self._decode_step_batched(B)
            done += n
        ev[2].record()
        outs = collect(done)
        # host waiting for the lazy requests' decodes is inside prefill_ms here: it is the time until all prompts are in
        self.last_timing = {"prompt_tokens": longest, "prefill_ms": ev[0].elapsed_time(ev[1]),
                            "decode_ms": ev[1].elapsed_time(ev[2]), "decode_steps": done - 1, "sequences": B}
        if not ignore_eos:
            outs = [seq[:next((i for i, t in enumerate(seq) if t in eos), len(seq))] for seq in outs]
        return [outs[slots[b]] if slots[b] is not None else errors[b] for b in range(n_req)]
