"""Row f2: Llama-3.2-11B-Vision ("mllama") on the same gfx950 kernel set - the model the
reference's Auditor falls back to (src/agents/vlm_auditor.py:81-83; request shape :152-158).

Everything arithmetic is a ``vis_*`` HIP entry point (hip.py); torch owns memory, streams and the hipGraph.
Published definition followed: transformers 5.15 ``models/mllama`` (``TF:``), see oracle/mllama_ref.py for the
CPU restatement the parity tests compare against.

Vision tower token order.  HF pads every tile from 1601 to 1608 tokens and masks only (padding x padding) pairs
(TF:modeling_mllama.py:75-99): real queries see every key, padding queries (pad rows and all rows of absent tiles)
see only real keys.  Attention is permutation-equivariant, so the tower runs on the order
    [ present tiles' 1601 tokens | absent tiles' 1601 tokens | the 7 pad rows of every tile ]
which makes both key sets contiguous ranges - two groups of plain work items for vis_attn_prefill - and leaves
the first max_tiles*1601 rows in exactly the order the cross-attention layers consume.

Scope: one image per prompt (what the reference sends).  Up to ``max_batch`` requests are in flight at once: every
request ("slot") owns its self-attention KV cache, its cross-attention keys/values and its position counter; prompt
passes run per request, then ONE decode loop serves all of them - every projection weight is streamed once per step
for the whole batch (gemm_decode_stream_kernel + skinny_finalize, the kernels of the Qwen2-VL batched decode), the
self-attention takes the sequence index on its grid and the cross-attention reads each sequence's own static keys
(vis_decode_cross_attn_batch).  Reference: VLMAuditorAgent.verify is called once per image of a batch
(src/agents/vlm_auditor.py:166-234, src/orchestration/graph.py:308-357).
"""
from __future__ import annotations

import math
import os
import threading
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import hip
from .mllama_weights import MllamaConfig, MllamaDeviceWeights


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


# ----------------------------------------------------------------------------- host logic (image geometry, rope)
def supported_aspect_ratios(max_tiles: int) -> List[Tuple[int, int]]:
    """TF:image_processing_pil_mllama.py get_all_supported_aspect_ratios."""
    return [(w, h) for w in range(1, max_tiles + 1) for h in range(1, max_tiles + 1) if w * h <= max_tiles]


def optimal_canvas(h: int, w: int, max_tiles: int, tile: int) -> Tuple[int, int]:
    """Canvas (height, width) in pixels: the smallest upscale >= 1 if one exists, else the largest downscale;
    ties go to the smallest area (TF:image_processing_pil_mllama.py get_optimal_tiled_canvas)."""
    best = None
    cands = []
    for (a, b) in supported_aspect_ratios(max_tiles):
        ch, cw = a * tile, b * tile
        sh, sw = ch / h, cw / w
        cands.append((sh if sw > sh else sw, ch, cw))
    ups = [c for c in cands if c[0] >= 1]
    sel = min(c[0] for c in ups) if ups else max(c[0] for c in cands if c[0] < 1)
    chosen = [c for c in cands if c[0] == sel]
    best = min(chosen, key=lambda c: c[1] * c[2]) if len(chosen) > 1 else chosen[0]
    # numpy's argmin takes the FIRST minimum; min() with a key does too
    return best[1], best[2]


def fit_to_canvas(h: int, w: int, ch: int, cw: int, tile: int) -> Tuple[int, int]:
    """TF:image_processing_pil_mllama.py get_image_size_fit_to_canvas."""
    tw = min(max(w, tile), cw)
    th = min(max(h, tile), ch)
    sh, sw = th / h, tw / w
    if sw < sh:
        return min(math.floor(h * sw) or 1, th), tw
    return th, min(math.floor(w * sh) or 1, tw)


def llama3_rope_tables(cfg: MllamaConfig, n: int) -> Tuple[np.ndarray, np.ndarray]:
    """cos/sin [n, head_dim] f32 for positions 0..n-1 with the llama3 frequency scaling
    (TF:modeling_rope_utils.py llama3; TF:modeling_mllama.py:747-758).  float32 arithmetic like the reference."""
    D = cfg.head_dim
    inv = (1.0 / (np.float32(cfg.rope_theta) ** (np.arange(0, D, 2, dtype=np.float32) / np.float32(D)))).astype(np.float32)
    if cfg.rope_factor and cfg.rope_factor > 0:
        low_wl = cfg.rope_orig_ctx / cfg.rope_low_freq
        high_wl = cfg.rope_orig_ctx / cfg.rope_high_freq
        wl = (np.float32(2 * math.pi) / inv).astype(np.float32)
        inv_l = np.where(wl > low_wl, inv / np.float32(cfg.rope_factor), inv).astype(np.float32)
        smooth = ((np.float32(cfg.rope_orig_ctx) / wl - np.float32(cfg.rope_low_freq)) /
                  np.float32(cfg.rope_high_freq - cfg.rope_low_freq)).astype(np.float32)
        smoothed = ((1 - smooth) * inv_l / np.float32(cfg.rope_factor) + smooth * inv_l).astype(np.float32)
        medium = ~(wl < high_wl) & ~(wl > low_wl)
        inv = np.where(medium, smoothed, inv_l).astype(np.float32)
    freqs = np.arange(n, dtype=np.float32)[:, None] * inv[None, :]
    emb = np.concatenate([freqs, freqs], axis=1)
    return np.cos(emb).astype(np.float32), np.sin(emb).astype(np.float32)


class MllamaEngine:
    """One mllama replica on one GPU.  Not re-entrant: callers serialise through ``self.lock``."""

    def __init__(self, cfg: MllamaConfig, weights: MllamaDeviceWeights, device, max_ctx: int = 4096, max_batch: int = 1):
        cfg.validate_for_kernels()
        hip.load()
        if not torch.cuda.is_available():
            raise hip.HipLibraryError("MllamaEngine needs a ROCm GPU (no CPU fallback exists)")
        self.cfg, self.w, self.device = cfg, weights, torch.device(device)
        self.max_ctx = _round_up(max_ctx, 64)
        self.lock = threading.Lock()
        dev, bf = self.device, torch.bfloat16
        Hq, Hkv, D, H = cfg.heads, cfg.kv_heads, cfg.head_dim, cfg.hidden
        self.self_idx = [i for i in range(cfg.layers) if i not in cfg.cross_layers]
        self.n_self, self.n_cross = len(self.self_idx), len(cfg.cross_layers)
        self.TP = cfg.max_tiles * cfg.tile_tokens
        self.Tk = _round_up(self.TP, 64)
        self.nsplit = max(1, -(-self.max_ctx // hip.DECODE_KEYS_PER_SPLIT))
        self.xsplit = -(-self.Tk // hip.DECODE_KEYS_PER_SPLIT)
        if not 1 <= max_batch <= 64:
            raise ValueError("max_batch must be in 1..64")
        Bm = self.max_batch = max_batch
        # per-request ("slot") state; slot 0 doubles as the single-sequence engine
        self.kcache_b = torch.zeros((Bm, self.n_self, Hkv, self.max_ctx, D), dtype=bf, device=dev)
        self.vcache_b = torch.zeros((Bm, self.n_self, Hkv, self.max_ctx, D), dtype=bf, device=dev)
        self.xk_b = torch.zeros((Bm, self.n_cross, Hkv, self.Tk, D), dtype=bf, device=dev)
        self.xv_b = torch.zeros((Bm, self.n_cross, Hkv, self.Tk, D), dtype=bf, device=dev)
        cos, sin = llama3_rope_tables(cfg, self.max_ctx)
        self.cos_t = torch.from_numpy(cos).to(dev)          # plain positions: one table for every sequence
        self.sin_t = torch.from_numpy(sin).to(dev)
        self.step_b = torch.zeros(Bm, dtype=torch.int32, device=dev)
        self.cur_b = torch.zeros(Bm, dtype=torch.int32, device=dev)
        self.nkeys_b = torch.zeros(Bm, dtype=torch.int32, device=dev)
        self.tokens_b = torch.zeros((Bm, self.max_ctx), dtype=torch.int32, device=dev)
        self.ws_val = torch.empty(max(256 * Bm, 2048), dtype=torch.float32, device=dev)
        self.ws_idx = torch.empty(max(256 * Bm, 2048), dtype=torch.int32, device=dev)
        self.logits_b = torch.empty((Bm, cfg.vocab), dtype=torch.float32, device=dev)
        self.kcache, self.vcache, self.xk, self.xv = self.kcache_b[0], self.vcache_b[0], self.xk_b[0], self.xv_b[0]
        self.step, self.cur_token, self.nkeys_m1 = self.step_b[0:1], self.cur_b[0:1], self.nkeys_b[0:1]
        self.tokens, self.logits = self.tokens_b[0], self.logits_b[0]
        nq = (Hq + 2 * Hkv) * D
        self.d_x = torch.empty((1, H), dtype=bf, device=dev)
        self.d_x2 = torch.empty((1, H), dtype=bf, device=dev)
        self.d_qkv = torch.empty(nq, dtype=bf, device=dev)
        self.d_attn = torch.empty(Hq * D, dtype=bf, device=dev)
        self.d_act = torch.empty(cfg.intermediate, dtype=bf, device=dev)
        ns = max(self.nsplit, self.xsplit)
        self.part_o = torch.empty(Bm * Hq * ns * D, dtype=torch.float32, device=dev)
        self.part_ml = torch.empty(Bm * Hq * ns * 2, dtype=torch.float32, device=dev)
        if Bm > 1:      # batched-decode activations and the stream-K partial workspace
            self.b_x = torch.empty((Bm, H), dtype=bf, device=dev)
            self.b_x2 = torch.empty((Bm, H), dtype=bf, device=dev)
            self.b_xn = torch.empty((Bm, H), dtype=bf, device=dev)
            self.b_xn2 = torch.empty((Bm, H), dtype=bf, device=dev)
            self.b_qkv = torch.empty((Bm, nq), dtype=bf, device=dev)
            self.b_attn = torch.empty((Bm, Hq * D), dtype=bf, device=dev)
            self.b_act = torch.empty((Bm, cfg.intermediate), dtype=bf, device=dev)
        # r05 experiment, OFF by default (VIS_DECODE_FUSED=1): every batched-decode projection as ONE launch (vis_decode_proj_bf16:
        # stream + split-K reduction + epilogue; Qwen2VLEngine.__init__ says why it is not the default).
        self.fused_proj = Bm > 1 and H % 128 == 0 and os.environ.get("VIS_DECODE_FUSED", "0") == "1"
        # batched decode: the self-attention launch finalises the qkv projection's partial slabs itself (vis_decode_attn_parts,
        # bit-identical to skinny_finalize + decode_attn); VIS_QKV_FOLD=0 keeps the two launches (A/B)
        self.fold_qkv = os.environ.get("VIS_QKV_FOLD", "1") == "1"
        if self.fused_proj:
            self.b_xw = torch.empty((Bm, H), dtype=bf, device=dev)
            self.b_x2w = torch.empty((Bm, H), dtype=bf, device=dev)
            self.b_ssq1 = torch.zeros((H // hip.SSQ_UNIT, hip.SSQ_LD), dtype=torch.float32, device=dev)
            self.b_ssq2 = torch.zeros((H // hip.SSQ_UNIT, hip.SSQ_LD), dtype=torch.float32, device=dev)
            lib = hip.load()
            need = max(int(lib.vis_decode_proj_ws_bytes(Bm, n, k, 0)) for n, k in
                       ((nq, H), (H, Hq * D), (2 * cfg.intermediate, H), (H, cfg.intermediate), (cfg.vocab, H)))
            if need <= 0:
                raise ValueError("vis_decode_proj_ws_bytes refused a projection shape of this model")
            self.b_proj_ws = torch.zeros(need, dtype=torch.uint8, device=dev)
        if Bm > 1:      # split-K slabs: the whole step with VIS_DECODE_FUSED=0, the long-K down projection at many sequences otherwise
            self.b_part = torch.empty(16 * hip.part_rows(Bm) * max(nq, H, 2 * cfg.intermediate), dtype=torch.float32,
                                      device=dev)
        # Single-sequence decode: the head of every SELF-attention layer (qkv projection -> rope / append / attention -> o
        # projection) as ONE launch, the lm_head with the pick's first stage in its epilogue - the Inspector's chained step
        # (csrc/decode_chain.hip, Qwen2VLEngine._decode_step; bit-identical to the launches it replaces); the eight
        # cross-attention layers keep their launches.  It runs while the context stays within what the device holds resident
        # for this head shape (11B: 768 projection + 32 merge workgroups + 8 attention items per 64 keys -> 1536 keys) and on
        # the separate launches beyond.  VIS_DECODE_CHAIN=0 = A/B.
        self.chain_sync: Optional[torch.Tensor] = None
        self.chain_ctx_limit = 0
        self._chain_launches = 0
        if os.environ.get("VIS_DECODE_CHAIN", "1") != "0" and hip.decode_chain_supported(Hq, Hkv, D, H):
            lim = hip.decode_chain_ctx_limit(Hq, Hkv, H)
            if lim > 0:
                self.chain_ws, self.chain_sync = hip.decode_chain_state(dev, Hq, Hkv, self.nsplit)
                self.chain_ctx_limit = lim
        self.slot_prompt_len = [0] * Bm
        self._graph: Optional[torch.cuda.CUDAGraph] = None
        self._graph_key = None
        self._graphs: Dict[tuple, torch.cuda.CUDAGraph] = {}
        self._graphs_b: Dict[tuple, torch.cuda.CUDAGraph] = {}
        self._vis_plans: Dict[tuple, "hip.AttnPlan"] = {}
        self.temperature, self.seed = 0.0, 0
        self.prompt_len = 0
        self._decoded = 0
        self.has_image = False
        self.decode_limit = 0

    # ------------------------------------------------------------------ preprocessing (geometry on host, pixels on GPU)
    def prepare_image(self, frame: torch.Tensor):
        """uint8 device frame [H, W, 3] -> (resized frame, tiles_h, tiles_w, aspect_ratio_id); bilinear resample on
        the GPU, bit-exact with the PIL call of the HF processor."""
        cfg = self.cfg
        h, w = int(frame.shape[0]), int(frame.shape[1])
        ch, cw = optimal_canvas(h, w, cfg.max_tiles, cfg.image_size)
        th, tw = ch // cfg.image_size, cw // cfg.image_size
        nh, nw = fit_to_canvas(h, w, ch, cw, cfg.image_size)
        if (nh, nw) != (h, w):
            frame = hip.resize_rgb(frame.contiguous(), nh, nw, kind="bilinear")
        ar_id = supported_aspect_ratios(cfg.max_tiles).index((th, tw)) + 1
        return frame.contiguous(), th, tw, ar_id

    # ------------------------------------------------------------------ vision tower
    def _vision_plan(self, layout: Sequence[Tuple[int, int, int]]) -> "hip.AttnPlan":
        """Attention plan of the tower over a stack of canvases [(first row, present rows, canvas rows)]: a present tile's
        rows attend every canvas row, the pad rows only the present ones (HF's mask as two key ranges).  The long segment
        of a canvas (51 row blocks per head for 2 x 2 tiles, on 48 resident slots per head) is key-split up to whole rounds
        of the chip (hip.plan_attn_items_split; r04 - DESIGN section 4 had it as "does not yet"); the rule looks at one
        canvas only, so an image's features do not depend on what it is stacked with.  Measured (r04, exact 11B shapes, same
        box, alternating): ONE canvas 44.2 -> 43.1 ms per prompt pass (two rounds with the second 6 % full become 2.02 rounds of
        mostly half items), FOUR stacked canvases (the batch path) 33.75 -> 34.0 ms per image (4.3 rounds of whole items already
        fill the chip; the halves add merge work) - and one rule has to serve both, or an image's features would differ between
        the single and the batched path.  Hence OFF by default (VIS_MLLAMA_ATTN_SPLIT=1 turns it on)."""
        key = tuple(layout)
        plan = self._vis_plans.get(key)
        if plan is None:
            segs = []
            for r0, n_real, n_all in layout:
                segs.append((r0, r0 + n_real, r0, r0 + n_all))
                if n_all > n_real:
                    segs.append((r0 + n_real, r0 + n_all, r0, r0 + n_real))
            if len(self._vis_plans) >= 16:
                self._vis_plans.clear()
            plan = self._vis_plans[key] = hip.make_vit_attn_plan(
                segs, self.device, self.cfg.v_heads,
                split=(self.cfg.v_head_dim == 80 and os.environ.get("VIS_MLLAMA_ATTN_SPLIT", "0") == "1"))
        return plan

    def vision_forward(self, frame: torch.Tensor, taps: Optional[dict] = None) -> Tuple[torch.Tensor, int]:
        """uint8 device frame [H, W, 3] -> (cross-attention states [max_tiles*tile_tokens, hidden] bf16, n_tiles)."""
        cfg, w, dev, bf = self.cfg, self.w, self.device, torch.bfloat16
        frame, th, tw, ar_id = self.prepare_image(frame)
        n_tiles = th * tw
        T, P, E, Hh, D = cfg.max_tiles, cfg.tile_tokens, cfg.v_hidden, cfg.v_heads, cfg.v_head_dim
        npad = (8 - P % 8) % 8
        TP, N, nR = T * P, T * (P + npad), n_tiles * P
        patches = torch.zeros((TP, w.patch_w.shape[1]), dtype=bf, device=dev)
        hip.patchify_tiles(frame, patches, th, tw, cfg.image_size, cfg.image_mean, cfg.image_std)
        x = torch.zeros((N, E), dtype=bf, device=dev)                      # pad rows start as exact zeros
        hip.gemm(patches, w.patch_w, residual=w.cls_pos[ar_id].view(TP, E), out=x[:TP])
        hip.layernorm(x[:TP], w.ln_pre_w, w.ln_pre_b, 1e-5, out=x[:TP])
        plan = self._vision_plan([(0, nR, N)])
        ld = _round_up(N, 64)
        y = torch.empty((N, E), dtype=bf, device=dev)
        qkv = torch.empty((N, 3 * E), dtype=bf, device=dev)
        q = torch.empty((Hh, N, D), dtype=bf, device=dev)
        k = torch.empty((Hh, N, D), dtype=bf, device=dev)
        vt = torch.empty((Hh, D, ld), dtype=bf, device=dev)
        att = torch.empty((N, E), dtype=bf, device=dev)
        hmid = torch.empty((N, cfg.v_mlp), dtype=bf, device=dev)
        feats = torch.empty((TP, cfg.v_out), dtype=bf, device=dev)
        scale = D ** -0.5

        def layer(b):
            hip.layernorm(x, b.ln1_w, b.ln1_b, cfg.v_eps, out=y)
            hip.gemm(y, b.qkv_w, out=qkv)
            hip.qkv_rope_split(qkv, None, None, q, k, None, vt, Hh, Hh, D)
            hip.attn_prefill_plan(q, k, vt, att, plan, scale)
            hip.gemm(att, b.o_w, residual=x, out=x)
            hip.layernorm(x, b.ln2_w, b.ln2_b, cfg.v_eps, out=y)
            hip.gemm(y, b.fc1_w, bias=b.fc1_b, act=hip.ACT_GELU_ERF, out=hmid)
            hip.gemm(hmid, b.fc2_w, bias=b.fc2_b, residual=x, out=x)

        j = 0
        for i, b in enumerate(w.v_layers):
            layer(b)
            if i in cfg.v_inter:
                feats[:, E * (1 + j):E * (2 + j)].copy_(x[:TP])
                j += 1
        hip.layernorm(x, w.ln_post_w, w.ln_post_b, 1e-5, out=x)
        tile_of = np.concatenate([np.repeat(np.arange(T), P), np.repeat(np.arange(T), npad)]).astype(np.int32)
        idx = hip.upload(ar_id * T + tile_of, dev)
        hip.add_rows(x, w.post_tile.view(-1, E), idx)
        for b in w.v_global:
            layer(b)
        feats[:, :E].copy_(x[:TP])
        if taps is not None:
            taps["vision_features"] = feats
        return hip.gemm(feats, w.proj_w, bias=w.proj_b), n_tiles

    def vision_forward_many(self, frames: Sequence[torch.Tensor]) -> List[Tuple[torch.Tensor, int]]:
        """The tower over SEVERAL requests' images in one pass (the batch seam: verify_many).  One image is 6432 rows:
        its projections are 1 - 2 rounds of the chip (qkv 390 tiles, fc1 520) and its attention list 816 workgroups for 768
        slots - two rounds, the second 6 % full; four images stacked make whole rounds of all of them.  Every image
        starts on a 64-row boundary (the attention kernel walks keys in absolute 64-row tiles), so its features - and with
        them the whole answer - are bit-identical to the single-image pass
        (test_batched_decode_matches_single_and_is_batch_invariant).  Row order inside an image as in vision_forward."""
        k = len(frames)
        if k == 1:
            return [self.vision_forward(frames[0])]
        cfg, w, dev, bf = self.cfg, self.w, self.device, torch.bfloat16
        T, P, E, Hh, D = cfg.max_tiles, cfg.tile_tokens, cfg.v_hidden, cfg.v_heads, cfg.v_head_dim
        npad = (8 - P % 8) % 8
        TP, N = T * P, T * (P + npad)
        NS = _round_up(N, 64)                         # rows per image in the stack
        x = torch.zeros((k * NS, E), dtype=bf, device=dev)
        layout, n_tiles_of = [], []
        tile_of = np.concatenate([np.repeat(np.arange(T), P), np.repeat(np.arange(T), npad),
                                  np.zeros(NS - N, dtype=np.int64)]).astype(np.int32)
        idx_all = np.empty(k * NS, dtype=np.int32)
        for i, frame in enumerate(frames):
            fr, th, tw, ar_id = self.prepare_image(frame)
            n_tiles = th * tw
            n_tiles_of.append(n_tiles)
            r0, nR = i * NS, n_tiles * P
            patches = torch.zeros((TP, w.patch_w.shape[1]), dtype=bf, device=dev)
            hip.patchify_tiles(fr, patches, th, tw, cfg.image_size, cfg.image_mean, cfg.image_std)
            hip.gemm(patches, w.patch_w, residual=w.cls_pos[ar_id].view(TP, E), out=x[r0:r0 + TP])
            hip.layernorm(x[r0:r0 + TP], w.ln_pre_w, w.ln_pre_b, 1e-5, out=x[r0:r0 + TP])
            layout.append((r0, nR, N))
            idx_all[r0:r0 + NS] = ar_id * T + tile_of
        plan = self._vision_plan(layout)
        M = k * NS
        y = torch.empty((M, E), dtype=bf, device=dev)
        qkv = torch.empty((M, 3 * E), dtype=bf, device=dev)
        q = torch.empty((Hh, M, D), dtype=bf, device=dev)
        kk = torch.empty((Hh, M, D), dtype=bf, device=dev)
        vt = torch.empty((Hh, D, M), dtype=bf, device=dev)
        att = torch.zeros((M, E), dtype=bf, device=dev)          # the NS - N tail rows of an image are never written
        hmid = torch.empty((M, cfg.v_mlp), dtype=bf, device=dev)
        feats = torch.empty((k, TP, cfg.v_out), dtype=bf, device=dev)
        scale = D ** -0.5

        def layer(b):
            hip.layernorm(x, b.ln1_w, b.ln1_b, cfg.v_eps, out=y)
            hip.gemm(y, b.qkv_w, out=qkv)
            hip.qkv_rope_split(qkv, None, None, q, kk, None, vt, Hh, Hh, D)
            hip.attn_prefill_plan(q, kk, vt, att, plan, scale)
            hip.gemm(att, b.o_w, residual=x, out=x)
            hip.layernorm(x, b.ln2_w, b.ln2_b, cfg.v_eps, out=y)
            hip.gemm(y, b.fc1_w, bias=b.fc1_b, act=hip.ACT_GELU_ERF, out=hmid)
            hip.gemm(hmid, b.fc2_w, bias=b.fc2_b, residual=x, out=x)

        def take(col0):
            for i in range(k):
                feats[i][:, col0:col0 + E].copy_(x[i * NS:i * NS + TP])

        j = 0
        for li, b in enumerate(w.v_layers):
            layer(b)
            if li in cfg.v_inter:
                take(E * (1 + j))
                j += 1
        hip.layernorm(x, w.ln_post_w, w.ln_post_b, 1e-5, out=x)
        hip.add_rows(x, w.post_tile.view(-1, E), hip.upload(idx_all, dev))
        for b in w.v_global:
            layer(b)
        take(0)
        cross = hip.gemm(feats.view(k * TP, cfg.v_out), w.proj_w, bias=w.proj_b)
        return [(cross[i * TP:(i + 1) * TP], n_tiles_of[i]) for i in range(k)]

    # ------------------------------------------------------------------ prefill
    def prefill(self, input_ids: Sequence[int], frame: Optional[torch.Tensor] = None, taps: Optional[dict] = None,
                temperature: float = 0.0, seed: int = 0, slot: int = 0,
                cross_states: Optional[Tuple[torch.Tensor, int]] = None) -> None:
        cfg, w, dev, bf = self.cfg, self.w, self.device, torch.bfloat16
        if not 0 <= slot < self.max_batch:
            raise ValueError("slot out of range")
        self.temperature, self.seed = float(temperature), int(seed)
        kcache, vcache, xk, xv = self.kcache_b[slot], self.vcache_b[slot], self.xk_b[slot], self.xv_b[slot]
        step, cur_token, nkeys_m1 = self.step_b[slot:slot + 1], self.cur_b[slot:slot + 1], self.nkeys_b[slot:slot + 1]
        tokens, logits = self.tokens_b[slot], self.logits_b[slot]
        S = len(input_ids)
        if S < 1 or S + 1 > self.max_ctx:
            raise ValueError(f"prompt of {S} tokens does not fit the context of {self.max_ctx}")
        ids_np = np.asarray(list(input_ids), dtype=np.int64)
        if ids_np.min() < 0 or ids_np.max() >= cfg.vocab + 8:
            raise ValueError("token id out of range")
        locs = np.nonzero(ids_np == cfg.image_token_id)[0]
        if len(locs) > 1:
            raise ValueError("one image per prompt")
        if (frame is None) != (len(locs) == 0):
            raise ValueError("image token and image frame must come together")
        H, Hq, Hkv, D = cfg.hidden, cfg.heads, cfg.kv_heads, cfg.head_dim
        P, TP = cfg.tile_tokens, self.TP
        has_image = frame is not None
        if slot == 0:
            self.has_image = has_image
        nm = int(locs[0]) if has_image else 0
        cross = None
        if has_image:
            # cross_states: the tower's output for this frame computed elsewhere (generate_batch stacks several requests' images)
            cross, n_tiles = cross_states if cross_states is not None else self.vision_forward(frame, taps)
            nR = n_tiles * P
            nkeys_m1.fill_(nR - 1)
            if taps is not None:
                taps["cross_states"] = cross
            xitems = [(q0, min(128, nm - q0), 0, TP) for q0 in range(0, nm, 128)] + \
                     [(q0, min(128, S - q0), 0, nR) for q0 in range(nm, S, 128)]
            xwork = torch.tensor(xitems, dtype=torch.int32, device=dev).reshape(-1, 4).contiguous()
            kvbuf = torch.empty((TP, 2 * Hkv * D), dtype=bf, device=dev)
            xvt = torch.empty((Hkv, D, self.Tk), dtype=bf, device=dev)
            q2 = torch.empty((S, Hq * D), dtype=bf, device=dev)
        x = torch.empty((S, H), dtype=bf, device=dev)
        hip.gather_rows(w.embed, hip.upload(ids_np.astype(np.int32), dev), x)
        cos, sin = self.cos_t[:S], self.sin_t[:S]
        work = hip.make_attn_pairs(0, S, dev)        # causal self-attention: paired query blocks (hip.attn_prefill_pairs)
        ld = _round_up(S, 64)
        nq = (Hq + 2 * Hkv) * D
        y = torch.empty((S, H), dtype=bf, device=dev)
        qkv = torch.empty((S, nq), dtype=bf, device=dev)
        q = torch.empty((Hq, S, D), dtype=bf, device=dev)
        vt = torch.empty((Hkv, D, ld), dtype=bf, device=dev)
        att = torch.empty((S, Hq * D), dtype=bf, device=dev)
        act = torch.empty((S, cfg.intermediate), dtype=bf, device=dev)
        scale = D ** -0.5
        si = ci = 0
        # Long-K projections at a prompt's few hundred rows (o: 3 x 16 tiles of 256^2 at K = 4096; down: the same tiles at
        # K = 14336) leave most of the chip idle as plain tiles: they run as K-slices (f32 slabs + fixed-order finalisation
        # with the residual, vis_gemm_bf16_splitk).  The slice count is a function of the LAYER shape only, never of the row
        # count, so a row's summation order - and slot / batch invariance - does not depend on the prompt (A/B: VIS_MLLAMA_SPLITK=0).
        ks_down = int(os.environ.get("VIS_MLLAMA_SPLITK", "4")) if cfg.intermediate >= 8192 else 0
        ks_o = int(os.environ.get("VIS_MLLAMA_SPLITK_O", "4")) if H >= 4096 else 0
        ks_qkv = int(os.environ.get("VIS_MLLAMA_SPLITK_QKV", "0"))
        swork = torch.empty(max(ks_down, ks_o, 1) * S * max(H, nq if ks_qkv else 0), dtype=torch.float32, device=dev) if (ks_down or ks_o or ks_qkv) else None

        def proj(a, wt, ks):        # x += a @ wt.T
            if ks >= 2:
                hip.gemm_splitk(a, wt, swork, ks, residual=x, out=x)
            else:
                hip.gemm(a, wt, residual=x, out=x)

        for li, lw in enumerate(w.layers):
            if lw.cross:
                if not has_image:
                    ci += 1
                    continue                      # text-only prompt: cross layers are skipped (TF:...:1128-1138)
                hip.gemm(cross, lw.kv_w, out=kvbuf)
                hip.rmsnorm_heads(kvbuf, lw.k_norm, Hkv, cfg.rms_eps)     # k_norm on the Hkv key heads of every row (one launch)
                hip.qkv_rope_split(kvbuf, None, None, None, xk[ci], xv[ci], xvt, 0, Hkv, D, k_pos0=0)
                hip.rmsnorm(x, lw.ln1_w, cfg.rms_eps, out=y)
                hip.gemm(y, lw.qkv_w, out=q2)
                hip.rmsnorm(q2.view(S * Hq, D), lw.q_norm, cfg.rms_eps, out=q2.view(S * Hq, D))
                hip.qkv_rope_split(q2, None, None, q, None, None, None, Hq, 0, D)
                hip.attn_prefill(q, xk[ci], xvt, att, xwork, False, scale)
                proj(att, lw.o_w, ks_o)
                keep = x[:nm].clone() if nm > 0 else None
                hip.rmsnorm(x, lw.ln2_w, cfg.rms_eps, out=y)
                hip.gemm(y, lw.gateup_w, act=hip.ACT_SWIGLU, out=act)
                proj(act, lw.down_w, ks_down)
                if keep is not None:              # rows before the image: MLP contribution zeroed (TF:...:697-699)
                    x[:nm].copy_(keep)
                ci += 1
            else:
                hip.rmsnorm(x, lw.ln1_w, cfg.rms_eps, out=y)
                if ks_qkv >= 2:
                    hip.gemm_splitk(y, lw.qkv_w, swork, ks_qkv, out=qkv)
                else:
                    hip.gemm(y, lw.qkv_w, out=qkv)
                hip.qkv_rope_split(qkv, cos, sin, q, kcache[si], vcache[si], vt, Hq, Hkv, D, k_pos0=0)
                hip.attn_prefill_pairs(q, kcache[si], vt, att, work, scale)
                proj(att, lw.o_w, ks_o)
                hip.rmsnorm(x, lw.ln2_w, cfg.rms_eps, out=y)
                hip.gemm(y, lw.gateup_w, act=hip.ACT_SWIGLU, out=act)
                proj(act, lw.down_w, ks_down)
                si += 1
            if taps is not None:
                taps[f"layer{li}"] = x.clone()
        hip.gemv(x[S - 1], w.lm_head, logits, norm_w=w.norm_w, eps=cfg.rms_eps)
        if taps is not None:
            taps["first_logits"] = logits.clone()
        step.fill_(S - 1)
        hip.argmax(logits, self.ws_val[256 * slot:256 * (slot + 1)], self.ws_idx[256 * slot:256 * (slot + 1)], tokens,
                   cur_token, step, self.temperature, self.seed + 0x9E3779B9 * slot)
        self.slot_prompt_len[slot] = S
        if slot == 0:
            self.prompt_len, self._decoded = S, 0
            self.decode_limit = self.max_ctx

    def _prefill_group(self, items: Sequence[tuple], temperature: float, seed: int) -> None:
        """The text decoder's prompt pass of SEVERAL requests with one prompt layout (verify_many: the same Auditor prompt in
        front of every image) over their stacked rows.  items: [(slot, ids, cross_states, n_tiles)], all prompts S tokens
        long with the image token at the same index.  At S ~ 700 a request is 2.75 row tiles of 256: stacked, four give 11
        (gate/up 112 x 3 = 1.3 rounds per request -> 4.8 for four).  Only the row-independent kernels see the stack
        (projections, norms); K / V projection of the image, rope / cache write and both attentions run per request.  The
        K order of every projection is M-independent and the K-slice counts are a function of the layer shape, so a
        request's tokens are those of its own pass (test_batched_decode_matches_single_and_is_batch_invariant)."""
        cfg, w, dev, bf = self.cfg, self.w, self.device, torch.bfloat16
        self.temperature, self.seed = float(temperature), int(seed)
        k = len(items)
        S = len(items[0][1])
        H, Hq, Hkv, D = cfg.hidden, cfg.heads, cfg.kv_heads, cfg.head_dim
        P, TP = cfg.tile_tokens, self.TP
        ids0 = np.asarray(list(items[0][1]), dtype=np.int64)
        nm = int(np.nonzero(ids0 == cfg.image_token_id)[0][0])
        M = k * S
        x = torch.empty((M, H), dtype=bf, device=dev)
        xworks, xvts = [], []
        for j, (slot, ids, cross, n_tiles) in enumerate(items):
            ids_np = np.asarray(list(ids), dtype=np.int64)
            locs = np.nonzero(ids_np == cfg.image_token_id)[0]
            if len(ids_np) != S or len(locs) != 1 or int(locs[0]) != nm:
                raise ValueError("_prefill_group: the prompts of a group must share length and image-token position")
            if ids_np.min() < 0 or ids_np.max() >= cfg.vocab + 8:
                raise ValueError("token id out of range")
            hip.gather_rows(w.embed, hip.upload(ids_np.astype(np.int32), dev), x[j * S:(j + 1) * S])
            nR = n_tiles * P
            self.nkeys_b[slot:slot + 1].fill_(nR - 1)
            xitems = [(q0, min(128, nm - q0), 0, TP) for q0 in range(0, nm, 128)] + \
                     [(q0, min(128, S - q0), 0, nR) for q0 in range(nm, S, 128)]
            xworks.append(xitems)
        xvt_all = torch.empty((k, Hkv, D, self.Tk), dtype=bf, device=dev)
        xvts = [xvt_all[j] for j in range(k)]
        xwork_all = torch.tensor(xworks, dtype=torch.int32, device=dev).reshape(k, -1, 4).contiguous()   # same item count: same S, nm
        xworks = [xwork_all[j] for j in range(k)]
        kvbuf = torch.empty((TP, 2 * Hkv * D), dtype=bf, device=dev)
        q2 = torch.empty((M, Hq * D), dtype=bf, device=dev)
        cos, sin = self.cos_t[:S], self.sin_t[:S]
        work = hip.make_attn_pairs(0, S, dev)
        ld = _round_up(S, 64)
        nq = (Hq + 2 * Hkv) * D
        y = torch.empty((M, H), dtype=bf, device=dev)
        qkv = torch.empty((M, nq), dtype=bf, device=dev)
        q = torch.empty((k, Hq, S, D), dtype=bf, device=dev)
        vt = torch.empty((k, Hkv, D, ld), dtype=bf, device=dev)
        att = torch.empty((M, Hq * D), dtype=bf, device=dev)
        act = torch.empty((M, cfg.intermediate), dtype=bf, device=dev)
        scale = D ** -0.5
        ks_down = int(os.environ.get("VIS_MLLAMA_SPLITK", "4")) if cfg.intermediate >= 8192 else 0
        ks_o = int(os.environ.get("VIS_MLLAMA_SPLITK_O", "4")) if H >= 4096 else 0
        ks_qkv = int(os.environ.get("VIS_MLLAMA_SPLITK_QKV", "0"))
        swork = torch.empty(max(ks_down, ks_o, 1) * M * max(H, nq if ks_qkv else 0), dtype=torch.float32, device=dev) \
            if (ks_down or ks_o or ks_qkv) else None

        def proj(a, wt, ks):        # x += a @ wt.T
            if ks >= 2:
                hip.gemm_splitk(a, wt, swork, ks, residual=x, out=x)
            else:
                hip.gemm(a, wt, residual=x, out=x)

        Tc = self.kcache_b.shape[3]
        many = os.environ.get("VIS_GROUP_ATTN", "1") == "1" and 1 < k <= hip.MAX_GROUP_REQUESTS and self.kcache_b.is_contiguous()
        si = ci = 0
        for lw in w.layers:
            if lw.cross:
                for j, (slot, _, cross, _) in enumerate(items):
                    hip.gemm(cross, lw.kv_w, out=kvbuf)
                    hip.rmsnorm_heads(kvbuf, lw.k_norm, Hkv, cfg.rms_eps)
                    hip.qkv_rope_split(kvbuf, None, None, None, self.xk_b[slot][ci], self.xv_b[slot][ci], xvts[j], 0, Hkv, D,
                                       k_pos0=0)
                hip.rmsnorm(x, lw.ln1_w, cfg.rms_eps, out=y)
                hip.gemm(y, lw.qkv_w, out=q2)
                hip.rmsnorm(q2.view(M * Hq, D), lw.q_norm, cfg.rms_eps, out=q2.view(M * Hq, D))
                if many and self.xk_b.is_contiguous():
                    xoff = [it[0] * self.xk_b.stride(0) + ci * self.xk_b.stride(1) for it in items]
                    for j in range(k):
                        hip.qkv_rope_split(q2[j * S:(j + 1) * S], None, None, q[j], None, None, None, Hq, 0, D)
                    hip.attn_prefill_many(q, self.xk_b, xvt_all, att, xwork_all, False, scale, xoff, self.Tk)
                else:
                    for j, (slot, _, _, _) in enumerate(items):
                        rows = slice(j * S, (j + 1) * S)
                        hip.qkv_rope_split(q2[rows], None, None, q[j], None, None, None, Hq, 0, D)
                        hip.attn_prefill(q[j], self.xk_b[slot][ci], xvts[j], att[rows], xworks[j], False, scale)
                proj(att, lw.o_w, ks_o)
                keep = [x[j * S:j * S + nm].clone() for j in range(k)] if nm > 0 else None
                hip.rmsnorm(x, lw.ln2_w, cfg.rms_eps, out=y)
                hip.gemm(y, lw.gateup_w, act=hip.ACT_SWIGLU, out=act)
                proj(act, lw.down_w, ks_down)
                if keep is not None:              # rows before the image: MLP contribution zeroed (TF:...:697-699)
                    for j in range(k):
                        x[j * S:j * S + nm].copy_(keep[j])
                ci += 1
            else:
                hip.rmsnorm(x, lw.ln1_w, cfg.rms_eps, out=y)
                if ks_qkv >= 2:
                    hip.gemm_splitk(y, lw.qkv_w, swork, ks_qkv, out=qkv)
                else:
                    hip.gemm(y, lw.qkv_w, out=qkv)
                if many:           # the group's requests as ONE launch each (per request bit-identical; see Qwen2VLEngine._prefill_group)
                    kv_off = [it[0] * self.kcache_b.stride(0) + si * self.kcache_b.stride(1) for it in items]
                    hip.qkv_rope_split_many(qkv, cos, sin, q, self.kcache_b, self.vcache_b, vt, Hq, Hkv, D, kv_off, Tc, k_pos0=0)
                    hip.attn_prefill_pairs_many(q, self.kcache_b, vt, att, work, scale, kv_off, Tc)
                else:
                    for j, (slot, _, _, _) in enumerate(items):
                        rows = slice(j * S, (j + 1) * S)
                        kc, vc = self.kcache_b[slot][si], self.vcache_b[slot][si]
                        hip.qkv_rope_split(qkv[rows], cos, sin, q[j], kc, vc, vt[j], Hq, Hkv, D, k_pos0=0)
                        hip.attn_prefill_pairs(q[j], kc, vt[j], att[rows], work, scale)
                proj(att, lw.o_w, ks_o)
                hip.rmsnorm(x, lw.ln2_w, cfg.rms_eps, out=y)
                hip.gemm(y, lw.gateup_w, act=hip.ACT_SWIGLU, out=act)
                proj(act, lw.down_w, ks_down)
                si += 1
        grouped = many and k <= 4 and H * 2 * (2 if k <= 2 else 4) <= 152 * 1024   # one pass over the lm_head for the group's last rows
        if grouped:
            lg = torch.empty((k, self.logits_b.shape[1]), dtype=torch.float32, device=dev)
            hip.gemv_rows(x[S - 1::S], w.lm_head, lg, norm_w=w.norm_w, eps=cfg.rms_eps)
        for j, (slot, _, _, _) in enumerate(items):
            logits = self.logits_b[slot]
            if grouped:
                logits.copy_(lg[j])
            else:
                hip.gemv(x[(j + 1) * S - 1], w.lm_head, logits, norm_w=w.norm_w, eps=cfg.rms_eps)
            self.step_b[slot:slot + 1].fill_(S - 1)
            hip.argmax(logits, self.ws_val[256 * slot:256 * (slot + 1)], self.ws_idx[256 * slot:256 * (slot + 1)],
                       self.tokens_b[slot], self.cur_b[slot:slot + 1], self.step_b[slot:slot + 1], self.temperature,
                       self.seed + 0x9E3779B9 * slot)
            self.slot_prompt_len[slot] = S
            if slot == 0:
                self.has_image = True
                self.prompt_len, self._decoded = S, 0
                self.decode_limit = self.max_ctx

    # ------------------------------------------------------------------ decode
    def _decode_step(self, chained: bool = False) -> None:
        cfg, w = self.cfg, self.w
        Hq, Hkv, D = cfg.heads, cfg.kv_heads, cfg.head_dim
        scale = D ** -0.5
        chained = chained and self.chain_sync is not None
        embed_in_chain = chained and not w.layers[0].cross      # the first layer's launch reads the embedding row itself
        if not embed_in_chain:
            hip.gather_rows(w.embed, self.cur_token, self.d_x)
        x, x2 = self.d_x, self.d_x2
        si = ci = 0
        for li, lw in enumerate(w.layers):
            if lw.cross:
                if not self.has_image:
                    ci += 1
                    continue
                dq = self.d_qkv[:Hq * D]
                hip.gemv(x[0], lw.qkv_w, dq, norm_w=lw.ln1_w, eps=cfg.rms_eps)
                hip.decode_cross_attn(dq, lw.q_norm, self.xk[ci], self.xv[ci], self.nkeys_m1, self.part_o,
                                      self.part_ml, self.d_attn, Hq, Hkv, D, self.xsplit, scale, cfg.rms_eps)
                ci += 1
                hip.gemv(self.d_attn, lw.o_w, x2[0], residual=x[0])
            elif chained:
                first = li == 0 and embed_in_chain
                hip.decode_chain(w.embed if first else x[0], lw.qkv_w, None, lw.ln1_w, lw.o_w, x2[0], self.cos_t, self.sin_t,
                                 self.kcache[si], self.vcache[si], self.step, self.chain_ws, self.chain_sync, Hq, Hkv, D,
                                 self.nsplit, scale, cfg.rms_eps, x_index=self.cur_token if first else None,
                                 ctx_bound=self.chain_ctx_limit)
                si += 1
            else:
                hip.gemv(x[0], lw.qkv_w, self.d_qkv, norm_w=lw.ln1_w, eps=cfg.rms_eps)
                hip.decode_attn(self.d_qkv, self.cos_t, self.sin_t, self.kcache[si], self.vcache[si], self.step,
                                self.part_o, self.part_ml, self.d_attn, Hq, Hkv, D, self.nsplit, scale)
                si += 1
                hip.gemv(self.d_attn, lw.o_w, x2[0], residual=x[0])
            hip.gemv(x2[0], lw.gateup_w, self.d_act, norm_w=lw.ln2_w, act=hip.ACT_SWIGLU, eps=cfg.rms_eps)
            hip.gemv(self.d_act, lw.down_w, x[0], residual=x2[0])
        if chained:     # the pick's first stage rides in the lm_head epilogue
            hip.gemv_argmax(x[0], w.lm_head, self.logits, self.ws_val, self.ws_idx, self.tokens, self.cur_token, self.step,
                            norm_w=w.norm_w, eps=cfg.rms_eps, temperature=self.temperature, seed=self.seed)
            return
        hip.gemv(x[0], w.lm_head, self.logits, norm_w=w.norm_w, eps=cfg.rms_eps)
        hip.argmax(self.logits, self.ws_val, self.ws_idx, self.tokens, self.cur_token, self.step, self.temperature,
                   self.seed)

    def _ensure_graph(self, chained: bool = False) -> torch.cuda.CUDAGraph:
        chained = chained and self.chain_sync is not None
        key = (self.temperature, self.seed, self.has_image, chained)
        if key in self._graphs:
            return self._graphs[key]
        snap = (self.step.clone(), self.cur_token.clone())
        s = torch.cuda.Stream(device=self.device)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self._decode_step(chained)               # warm-up outside capture
        torch.cuda.current_stream().wait_stream(s)
        self.step.copy_(snap[0]); self.cur_token.copy_(snap[1])
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):     # another agent's thread may be allocating
            self._decode_step(chained)
        self.step.copy_(snap[0]); self.cur_token.copy_(snap[1])
        if len(self._graphs) >= 6:
            self._graphs.pop(next(iter(self._graphs)))
        self._graphs[key] = g
        return g

    def _decode_steps(self, n_steps: int, use_graph: bool, chained: bool) -> None:
        if use_graph:
            g = self._ensure_graph(chained)
            for _ in range(n_steps):
                g.replay()
        else:
            for _ in range(n_steps):
                self._decode_step(chained)

    def decode(self, n_steps: int, use_graph: bool = True) -> None:
        if self.prompt_len + self._decoded + n_steps >= self.decode_limit:
            raise ValueError("decode would run past the context window")
        base = self.prompt_len + self._decoded       # step k of this call sees base + k cached keys
        n_chain = max(0, min(n_steps, self.chain_ctx_limit - base)) if self.chain_sync is not None else 0
        if n_chain:
            # chained launches wait inside the grid for workgroups of the SAME launch: one at a time per device, whatever
            # engine, thread or stream it comes from (the Inspector's engine shares the GPU): ordered with an event, as in
            # Qwen2VLEngine.decode
            from .engine import _CHAIN_LAST, _chain_order_lock
            n_self = sum(1 for lw in self.w.layers if not lw.cross)
            self._chain_launches += n_chain * n_self
            if self._chain_launches >= (1 << 31):     # long before the 32-bit launch counter can wrap (Qwen2VLEngine._chain_epoch_guard)
                self.chain_sync.zero_()
                self.chain_ws.zero_()
                self._chain_launches = n_chain * n_self
            with _chain_order_lock(self.device.index):
                cur = torch.cuda.current_stream(self.device)
                prev = _CHAIN_LAST.get(self.device.index)
                if prev is not None:
                    cur.wait_event(prev)
                self._decode_steps(n_chain, use_graph, True)
                ev = torch.cuda.Event()
                ev.record(cur)
                _CHAIN_LAST[self.device.index] = ev
        if n_steps > n_chain:
            self._decode_steps(n_steps - n_chain, use_graph, False)
        self._decoded += n_steps

    def check_chain(self) -> None:
        """Raise hip.ChainStalled if a bounded wait inside a chained launch gave up (results invalid); see Qwen2VLEngine.check_chain."""
        if self.chain_sync is not None and int(self.chain_sync[hip.CHAIN_STATUS_WORD].item()) != 0:
            self.chain_sync.zero_()
            self.chain_ws.zero_()
            self._chain_launches = 0
            raise hip.ChainStalled("vis_decode_chain: a hand-off wait inside the launch timed out (decode results invalid)")

    # ---- batched decode: B in-flight requests (slots 0..B-1, all with an image) share every weight read of a step
    def _decode_step_batched(self, B: int) -> None:
        """Every projection = gemm_decode (weights streamed once for all B sequences, stream-K f32 partials) +
        skinny_finalize (row-wise: sum, residual / SwiGLU, and the RMSNorm of the NEXT projection); self-attention and
        cross-attention take the sequence index on the grid."""
        if self.fused_proj:
            return self._decode_step_fused(B)
        cfg, w = self.cfg, self.w
        Hq, Hkv, D, H = cfg.heads, cfg.kv_heads, cfg.head_dim, cfg.hidden
        scale, eps = D ** -0.5, cfg.rms_eps
        x, x2, xn, xn2 = self.b_x[:B], self.b_x2[:B], self.b_xn[:B], self.b_xn2[:B]
        qkv, att, act, part = self.b_qkv[:B], self.b_attn[:B], self.b_act[:B], self.b_part
        nq = qkv.shape[1]
        cosb = self.cos_t.unsqueeze(0).expand(B, -1, -1)          # batch stride 0: the rope table is shared
        sinb = self.sin_t.unsqueeze(0).expand(B, -1, -1)
        hip.gather_rows(w.embed, self.cur_b[:B], x)
        hip.rmsnorm(x, w.layers[0].ln1_w, eps, out=xn)
        n_layers = len(w.layers)
        si = ci = 0
        for li, lw in enumerate(w.layers):
            if lw.cross:
                q = qkv[:, :Hq * D]
                ks = hip.decode_gemm(xn, lw.qkv_w, part=part)
                hip.skinny_finalize(part, ks, q, Hq * D, eps=eps)
                hip.decode_cross_attn_batch(q, lw.q_norm, self.xk_b[:B, ci], self.xv_b[:B, ci], self.nkeys_b[:B],
                                            self.part_o, self.part_ml, att, Hq, Hkv, D, self.xsplit, scale, eps)
                ci += 1
            else:
                ks = hip.decode_gemm(xn, lw.qkv_w, part=part)
                if self.fold_qkv:      # the attention workgroups finalise the qkv columns they read (same bits, one launch less)
                    hip.decode_attn_parts(part, ks, cosb, sinb, self.kcache_b[:B, si], self.vcache_b[:B, si], self.step_b[:B],
                                          self.part_o, self.part_ml, att, Hq, Hkv, D, self.nsplit, scale)
                else:
                    hip.skinny_finalize(part, ks, qkv, nq, eps=eps)
                    hip.decode_attn(qkv, cosb, sinb, self.kcache_b[:B, si], self.vcache_b[:B, si], self.step_b[:B],
                                    self.part_o, self.part_ml, att, Hq, Hkv, D, self.nsplit, scale)
                si += 1
            ks = hip.decode_gemm(att, lw.o_w, part=part)
            hip.skinny_finalize(part, ks, x2, H, residual=x, norm_w=lw.ln2_w, yn=xn2, eps=eps)
            ks = hip.decode_gemm(xn2, lw.gateup_w, part=part)
            hip.skinny_finalize(part, ks, act, 2 * cfg.intermediate, swiglu=True, eps=eps)
            ks = hip.decode_gemm(act, lw.down_w, part=part)
            next_norm = w.layers[li + 1].ln1_w if li + 1 < n_layers else w.norm_w
            hip.skinny_finalize(part, ks, x, H, residual=x2, norm_w=next_norm, yn=xn, eps=eps)
        hip.decode_gemm(xn, w.lm_head, out=self.logits_b[:B])
        hip.argmax(self.logits_b[:B], self.ws_val, self.ws_idx, self.tokens_b[:B], self.cur_b[:B], self.step_b[:B],
                   self.temperature, self.seed)

    def _decode_step_fused(self, B: int) -> None:
        """The batched step with every projection as ONE launch (r05, csrc/decode_stream.hip; see Qwen2VLEngine._decode_step_fused):
        q / qkv (plain, rs of the input norm), o (+ residual, x ln2_w, sums of squares), gate/up (SwiGLU, rs), down (+ residual,
        x the next ln1_w, sums of squares); 5 launches per layer instead of 9.  The tanh gates of the cross-attention layers are
        folded into o_w / down_w at load time, so both layer kinds share the sequence."""
        cfg, w = self.cfg, self.w
        Hq, Hkv, D, H = cfg.heads, cfg.kv_heads, cfg.head_dim, cfg.hidden
        scale, eps = D ** -0.5, cfg.rms_eps
        x, x2, xw, x2w = self.b_x[:B], self.b_x2[:B], self.b_xw[:B], self.b_x2w[:B]
        qkv, att, act = self.b_qkv[:B], self.b_attn[:B], self.b_act[:B]
        s1, s2, ws = self.b_ssq1, self.b_ssq2, self.b_proj_ws
        cosb = self.cos_t.unsqueeze(0).expand(B, -1, -1)          # batch stride 0: the rope table is shared
        sinb = self.sin_t.unsqueeze(0).expand(B, -1, -1)
        # the long-K down projection at many sequences keeps the r02-r04 pair of launches (Qwen2VLEngine._decode_step_fused)
        down_pair = os.environ.get("VIS_DOWN_PAIR", "1") != "0" and \
            hip.decode_proj_form(B, H, cfg.intermediate, hip.DP_RESID_NORMW, False, False) == "streamk"
        s1_in = None if down_pair else s1
        if down_pair:
            hip.gather_rows(w.embed, self.cur_b[:B], x)
            hip.rmsnorm(x, w.layers[0].ln1_w, eps, out=xw)
        else:
            hip.decode_prep_rows(w.embed, self.cur_b[:B], w.layers[0].ln1_w, x, xw, s1)
        n_layers = len(w.layers)
        si = ci = 0
        for li, lw in enumerate(w.layers):
            if lw.cross:
                q = qkv[:, :Hq * D]
                hip.decode_proj(xw, lw.qkv_w, ws, hip.DP_PLAIN, out=q, ssq_in=s1_in, norm_dim=H, eps=eps)
                hip.decode_cross_attn_batch(q, lw.q_norm, self.xk_b[:B, ci], self.xv_b[:B, ci], self.nkeys_b[:B],
                                            self.part_o, self.part_ml, att, Hq, Hkv, D, self.xsplit, scale, eps)
                ci += 1
            else:
                hip.decode_proj(xw, lw.qkv_w, ws, hip.DP_PLAIN, out=qkv, ssq_in=s1_in, norm_dim=H, eps=eps)
                hip.decode_attn(qkv, cosb, sinb, self.kcache_b[:B, si], self.vcache_b[:B, si], self.step_b[:B],
                                self.part_o, self.part_ml, att, Hq, Hkv, D, self.nsplit, scale)
                si += 1
            hip.decode_proj(att, lw.o_w, ws, hip.DP_RESID_NORMW, out=x2, out_w=x2w, residual=x, norm_w=lw.ln2_w, ssq_out=s2)
            hip.decode_proj(x2w, lw.gateup_w, ws, hip.DP_SWIGLU, out=act, ssq_in=s2, norm_dim=H, eps=eps)
            next_norm = w.layers[li + 1].ln1_w if li + 1 < n_layers else w.norm_w
            if down_pair:
                ks = hip.decode_gemm(act, lw.down_w, part=self.b_part)
                hip.skinny_finalize(self.b_part, ks, x, H, residual=x2, norm_w=next_norm, yn=xw, eps=eps)
            else:
                hip.decode_proj(act, lw.down_w, ws, hip.DP_RESID_NORMW, out=x, out_w=xw, residual=x2, norm_w=next_norm,
                                ssq_out=s1)
        hip.decode_proj(xw, w.lm_head, ws, hip.DP_PLAIN, out=self.logits_b[:B], ssq_in=s1_in, norm_dim=H, eps=eps)
        hip.argmax(self.logits_b[:B], self.ws_val, self.ws_idx, self.tokens_b[:B], self.cur_b[:B], self.step_b[:B],
                   self.temperature, self.seed)

    def _ensure_graph_batched(self, B: int) -> torch.cuda.CUDAGraph:
        key = (self.temperature, self.seed, B)
        if key in self._graphs_b:
            return self._graphs_b[key]
        snap = (self.step_b.clone(), self.cur_b.clone())
        s = torch.cuda.Stream(device=self.device)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self._decode_step_batched(B)             # warm-up outside capture
        torch.cuda.current_stream().wait_stream(s)
        self.step_b.copy_(snap[0]); self.cur_b.copy_(snap[1])
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):     # another agent's thread may be allocating
            self._decode_step_batched(B)
        self.step_b.copy_(snap[0]); self.cur_b.copy_(snap[1])
        if len(self._graphs_b) >= 4:
            self._graphs_b.pop(next(iter(self._graphs_b)))
        self._graphs_b[key] = g
        return g

    def generate_batch(self, requests: Sequence, max_new_tokens: int = 128,
                       temperature: float = 0.0, seed: int = 0, stop_on_eos: bool = True, use_graph: bool = True,
                       chunk: int = 16) -> list:
        """requests: [(input_ids, frame)] for up to max_batch images (every request carries an image: the batched step
        always runs the cross-attention layers).  Prompt passes run per request; the decode steps are shared.
        A request may be a zero-argument callable returning the pair (the batch seam: it waits for the image's host
        decode, so the prompt pass of image 0 runs while images 1.. are still being decoded); one that raises gets no
        slot and its exception takes its place in the returned list."""
        n_req = len(requests)
        if not 1 <= n_req <= self.max_batch:
            raise ValueError(f"batch of {n_req} does not fit max_batch={self.max_batch}")
        lazy = any(callable(r) for r in requests)
        if lazy and n_req == 1:
            # one lazy request (always the case with max_batch == 1, where the batched buffers do not even exist): the
            # single-sequence path; its failure stays its own, as in the batched form
            try:
                ids, fr = requests[0]() if callable(requests[0]) else requests[0]
                return [self.generate(ids, fr, max_new_tokens, temperature, seed, stop_on_eos, use_graph)]
            except Exception as e:      # noqa: BLE001
                return [e]
        if not lazy and (n_req == 1 or any(fr is None for _, fr in requests)):
            if n_req > 1:
                raise ValueError("generate_batch needs an image in every request (text-only prompts go through generate)")
            ids, fr = requests[0]
            return [self.generate(ids, fr, max_new_tokens, temperature, seed, stop_on_eos, use_graph)]
        slots: List[Optional[int]] = [None] * n_req
        errors: List[Optional[Exception]] = [None] * n_req
        B = 0
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record()
        # The prompt passes are independent kernel chains (own buffers, own cache slot): issued round-robin on two HIP
        # streams (VIS_PREFILL_STREAMS, as in Qwen2VLEngine.prefill_many) the ragged last round of one image's GEMM /
        # attention grids - the 6432-row tower is 1-1.5 rounds of the chip per projection - is filled by the other's.
        n_streams = max(1, min(n_req, int(os.environ.get("VIS_PREFILL_STREAMS", "2"))))
        cur = torch.cuda.current_stream(self.device)
        if n_streams > 1 and len(getattr(self, "_prefill_streams", [])) < n_streams:
            self._prefill_streams = [torch.cuda.Stream(device=self.device) for _ in range(n_streams)]
        vb = max(1, int(os.environ.get("VIS_VIT_BATCH", "4"))) if n_streams > 1 else 1
        for g0 in range(0, n_req, vb):
            grp = []                                     # (request index, ids, frame) of the requests that resolved
            for b in range(g0, min(n_req, g0 + vb)):
                r = requests[b]
                try:
                    ids, fr = r() if callable(r) else r
                    if fr is None:
                        raise ValueError("generate_batch needs an image in every request")
                    grp.append((b, ids, fr))
                except Exception as e:      # noqa: BLE001 - a lazy request's failure stays its own
                    if not lazy:
                        raise
                    errors[b] = e
            if not grp:
                continue
            try:      # the tower once over the group's images (whole rounds of the chip), on the current stream
                crosses = self.vision_forward_many([fr for _, _, fr in grp]) if len(grp) > 1 else [None] * len(grp)
            except Exception as e:      # noqa: BLE001
                if not lazy:
                    raise
                for b, _, _ in grp:
                    errors[b] = e
                continue
            # requests with one prompt layout (the Auditor's fixed prompt): ONE stacked text pass (VIS_MERGE_PREFILL=0: A/B)
            stack = len(grp) > 1 and n_streams > 1 and os.environ.get("VIS_MERGE_PREFILL", "1") != "0"
            if stack:
                l0 = [i for i, t in enumerate(grp[0][1]) if t == self.cfg.image_token_id]
                stack = all(len(ids) == len(grp[0][1]) and
                            [i for i, t in enumerate(ids) if t == self.cfg.image_token_id] == l0 for _, ids, _ in grp) and len(l0) == 1
            if stack:
                st = self._prefill_streams[(g0 // vb) % n_streams]      # consecutive groups alternate streams
                st.wait_stream(cur)
                try:
                    with torch.cuda.stream(st):
                        items = []
                        for (b, ids, fr), cs in zip(grp, crosses):
                            cs[0].record_stream(st)
                            items.append((B + len(items), ids, cs[0], cs[1]))
                        self._prefill_group(items, temperature, seed)
                    for (b, _, _) in grp:
                        slots[b] = B
                        B += 1
                except Exception as e:      # noqa: BLE001
                    if not lazy:
                        raise
                    for b, _, _ in grp:
                        errors[b] = e
                continue
            for (b, ids, fr), cs in zip(grp, crosses):
                try:
                    if n_streams > 1:
                        st = self._prefill_streams[B % n_streams]
                        st.wait_stream(cur)          # the frame's upload / JPEG kernels and the group's tower ran on `cur`
                        with torch.cuda.stream(st):
                            fr.record_stream(st)
                            if cs is not None:
                                cs[0].record_stream(st)
                            self.prefill(ids, fr, temperature=temperature, seed=seed, slot=B, cross_states=cs)
                    else:
                        self.prefill(ids, fr, temperature=temperature, seed=seed, slot=B, cross_states=cs)
                except Exception as e:      # noqa: BLE001
                    if not lazy:
                        raise
                    errors[b] = e
                    continue
                slots[b] = B
                B += 1
        if n_streams > 1:
            for st in self._prefill_streams[:n_streams]:
                cur.wait_stream(st)
        ev[1].record()
        if B == 0:
            return list(errors)
        longest = max(self.slot_prompt_len[s] for s in range(B))
        max_new_tokens = max(1, min(max_new_tokens, self.max_ctx - longest - 1))
        eos = set(self.cfg.eos_ids)
        starts = [self.slot_prompt_len[b] - 1 for b in range(B)]

        def collect(n):
            t = self.tokens_b[:B].cpu()
            return [t[b, starts[b]:starts[b] + n].tolist() for b in range(B)]

        done = 1
        g = self._ensure_graph_batched(B) if use_graph else None
        while done < max_new_tokens:
            if stop_on_eos and all(any(t in eos for t in seq) for seq in collect(done)):
                break
            n = min(chunk if stop_on_eos else max_new_tokens, max_new_tokens - done)
            for _ in range(n):
                if g is not None:
                    g.replay()
                else:
                    self._decode_step_batched(B)
            done += n
        ev[2].record()
        outs = collect(done)
        self.last_timing = {"prompt_tokens": longest, "prefill_ms": ev[0].elapsed_time(ev[1]),
                            "decode_ms": ev[1].elapsed_time(ev[2]), "decode_steps": done - 1, "sequences": B}
        if stop_on_eos:
            outs = [seq[:next((i + 1 for i, t in enumerate(seq) if t in eos), len(seq))] for seq in outs]
        return [outs[slots[b]] if slots[b] is not None else errors[b] for b in range(n_req)]

    def generated(self, n: int) -> List[int]:
        s = self.prompt_len - 1          # the token generated at step i is stored at index (its position - 1)
        toks = self.tokens[s:s + n].cpu().tolist()
        self.check_chain()
        return toks

    def generate(self, input_ids: Sequence[int], frame: Optional[torch.Tensor] = None, max_new_tokens: int = 128,
                 temperature: float = 0.0, seed: int = 0, stop_on_eos: bool = True, use_graph: bool = True,
                 chunk: int = 32) -> List[int]:
        try:
            return self._generate(input_ids, frame, max_new_tokens, temperature, seed, stop_on_eos, use_graph, chunk)
        except hip.ChainStalled as e:
            # a chained launch could not get its waiting workgroups resident (another process on the GPU): the request is served
            # again on the separate launches - identical tokens - and this engine stays on them
            import logging
            logging.getLogger("vision_inspection_system_amd.engine").warning("%s - continuing on the unchained decode step", e)
            self.chain_sync = None
            self._graphs.clear()
            return self._generate(input_ids, frame, max_new_tokens, temperature, seed, stop_on_eos, use_graph, chunk)

    def _generate(self, input_ids, frame, max_new_tokens, temperature, seed, stop_on_eos, use_graph, chunk) -> List[int]:
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]      # per-stage device time, as in Qwen2VLEngine
        ev[0].record()
        self.prefill(input_ids, frame, temperature=temperature, seed=seed)
        ev[1].record()
        max_new_tokens = min(max_new_tokens, self.max_ctx - len(input_ids) - 1)
        eos = set(self.cfg.eos_ids)
        done = 1
        while done < max_new_tokens:
            toks = self.generated(done)
            if stop_on_eos and any(t in eos for t in toks):
                break
            n = min(chunk, max_new_tokens - done)
            self.decode(n, use_graph)
            done += n
        ev[2].record()
        toks = self.generated(done)                      # D2H: synchronises, the events have completed
        self.last_timing = {"prompt_tokens": len(input_ids), "prefill_ms": ev[0].elapsed_time(ev[1]),
                            "decode_ms": ev[1].elapsed_time(ev[2]), "decode_steps": done - 1, "sequences": 1}
        if stop_on_eos:
            for i, t in enumerate(toks):
                if t in eos:
                    return toks[:i + 1]
        return toks
