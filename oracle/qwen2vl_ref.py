"""ORACLE (test infrastructure - never imported by the product path).

CPU restatement, in plain fp32 PyTorch tensor algebra, of the arithmetic the
reference's remote endpoint performs for one ``chat.completions.create`` call
(reference call sites: src/agents/vlm_inspector.py:105-111,
src/agents/vlm_auditor.py:117-129,:152-158).  The reference itself contains no
model arithmetic, so the algorithm restated here is the
published Qwen2-VL definition as shipped in transformers 5.15.0
(``TF:`` = transformers/models/qwen2_vl/):

  smart_resize / patchify      TF:image_processing_pil_qwen2_vl.py:57-84,:156-190
  patch embed (Conv3d == GEMM) TF:modeling_qwen2_vl.py:251-274
  ViT block, 2-D rope, varlen  TF:modeling_qwen2_vl.py:225-248,:342-451
  patch merger                 TF:modeling_qwen2_vl.py:277-291
  M-RoPE ids / cos,sin / apply TF:modeling_qwen2_vl.py:156-222,:914-1016
  decoder layer, RMSNorm, MLP  TF:modeling_qwen2_vl.py:96-110,:453-466,:469-624
  image-token scatter          TF:modeling_qwen2_vl.py:1052-1100,:1144-1200

Pinning: tests/test_oracle.py checks this file against golden vectors produced
by the real transformers modules (tests/golden/gen_qwen2vl_golden.py ->
tests/golden/qwen2vl_tiny.npz, smart_resize.json): per-stage activations, logits and
16 greedy tokens of a tiny seeded config, single- and two-image (varlen) cases.
Logit-level parity with the reference's *remote service* is unpinned by
construction (no weights, no endpoint offline) - see DESIGN.md.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


@dataclass
class RefConfig:
    # text
    hidden: int = 3584
    layers: int = 28
    heads: int = 28
    kv_heads: int = 4
    intermediate: int = 18944
    vocab: int = 152064
    rms_eps: float = 1e-6
    rope_theta: float = 1e6
    mrope_section: Tuple[int, int, int] = (16, 24, 24)
    # vision
    v_depth: int = 32
    v_embed: int = 1280
    v_heads: int = 16
    v_mlp: int = 5120
    patch: int = 14
    temporal: int = 2
    merge: int = 2
    image_token_id: int = 151655

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads

    @property
    def v_head_dim(self) -> int:
        return self.v_embed // self.v_heads


# ----------------------------------------------------------------------------- preprocessing
def smart_resize(height: int, width: int, factor: int = 28, min_pixels: int = 56 * 56,
                 max_pixels: int = 28 * 28 * 1280) -> Tuple[int, int]:
    if max(height, width) / min(height, width) > 200:
        raise ValueError("absolute aspect ratio must be smaller than 200")
    h_bar = round(height / factor) * factor
    w_bar = round(width / factor) * factor
    if h_bar * w_bar > max_pixels:
        beta = math.sqrt((height * width) / max_pixels)
        h_bar = max(factor, math.floor(height / beta / factor) * factor)
        w_bar = max(factor, math.floor(width / beta / factor) * factor)
    elif h_bar * w_bar < min_pixels:
        beta = math.sqrt(min_pixels / (height * width))
        h_bar = math.ceil(height * beta / factor) * factor
        w_bar = math.ceil(width * beta / factor) * factor
    return h_bar, w_bar


def patchify(img_chw: np.ndarray, patch: int = 14, merge: int = 2, temporal: int = 2) -> Tuple[np.ndarray, int, int]:
    """[C,H,W] float32 (already normalised) -> ([gh*gw, C*temporal*patch*patch], gh, gw)."""
    c, h, w = img_chw.shape
    gh, gw = h // patch, w // patch
    p = img_chw.reshape(c, gh // merge, merge, patch, gw // merge, merge, patch)
    p = np.transpose(p, (1, 4, 2, 5, 0, 3, 6))
    p = np.broadcast_to(p[:, :, :, :, :, None, :, :], (*p.shape[:5], temporal, *p.shape[5:]))
    return np.ascontiguousarray(p.reshape(gh * gw, c * temporal * patch * patch)), gh, gw


def preprocess_u8(img_hwc_u8: np.ndarray, patch: int = 14, merge: int = 2, temporal: int = 2):
    """Resized uint8 RGB frame [H,W,3] (H,W multiples of 28) -> pixel_values rows, (1, gh, gw)."""
    x = img_hwc_u8.astype(np.float32) * np.float32(1.0 / 255.0)
    x = (x - np.array(CLIP_MEAN, np.float32)) / np.array(CLIP_STD, np.float32)
    pv, gh, gw = patchify(np.ascontiguousarray(x.transpose(2, 0, 1)), patch, merge, temporal)
    return pv, (1, gh, gw)


# ----------------------------------------------------------------------------- shared pieces
def rotate_half(x: torch.Tensor) -> torch.Tensor:
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def rms_norm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    var = x.pow(2).mean(-1, keepdim=True)
    return w * (x * torch.rsqrt(var + eps))


# ----------------------------------------------------------------------------- vision tower
def vision_pos_ids(grids: Sequence[Tuple[int, int, int]], merge: int) -> torch.Tensor:
    """(h, w) index of every patch in 2x2-merge-block order: [N, 2]."""
    out = []
    for (t, h, w) in grids:
        hp = torch.arange(h).unsqueeze(1).expand(h, w)
        wp = torch.arange(w).unsqueeze(0).expand(h, w)
        shape = (h // merge, merge, w // merge, merge)
        hp = hp.reshape(shape).transpose(1, 2).flatten()
        wp = wp.reshape(shape).transpose(1, 2).flatten()
        out.append(torch.stack([hp, wp], dim=-1).repeat(t, 1))
    return torch.cat(out, dim=0)


def vision_cos_sin(cfg: RefConfig, grids) -> Tuple[torch.Tensor, torch.Tensor]:
    dim = cfg.v_head_dim // 2
    inv_freq = 1.0 / (10000.0 ** (torch.arange(0, dim, 2, dtype=torch.float32) / dim))
    pos = vision_pos_ids(grids, cfg.merge).float()
    freqs = (pos.unsqueeze(-1) * inv_freq).flatten(1)  # [N, head_dim/2]
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos(), emb.sin()


def vision_forward(cfg: RefConfig, sd: Dict[str, torch.Tensor], pixel_values: torch.Tensor, grids,
                   taps: Optional[dict] = None, act_fp8: bool = False) -> torch.Tensor:
    """pixel_values [N, 1176] -> merged image embeddings [N/4, hidden].  act_fp8: fake-quantise the input of the four
    block projections per row to e4m3 (fp8 configuration; ``sd`` then holds the de-quantised e4m3 block weights)."""
    E, H, D = cfg.v_embed, cfg.v_heads, cfg.v_head_dim
    fq = fake_quant_rows_e4m3 if act_fp8 else (lambda t: t)
    w_pe = sd["visual.patch_embed.proj.weight"].reshape(E, -1)
    x = pixel_values @ w_pe.t()
    if taps is not None:
        taps["patch_embed"] = x.clone()
    cos, sin = vision_cos_sin(cfg, grids)
    seg, start = [], 0
    for (t, h, w) in grids:
        for _ in range(t):
            seg.append((start, start + h * w))
            start += h * w
    for i in range(cfg.v_depth):
        p = f"visual.blocks.{i}."
        y = fq(F.layer_norm(x, (E,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-6))
        qkv = y @ sd[p + "attn.qkv.weight"].t() + sd[p + "attn.qkv.bias"]
        qkv = qkv.reshape(-1, 3, H, D)
        q, k, v = qkv[:, 0], qkv[:, 1], qkv[:, 2]  # [N, H, D]
        q = q * cos[:, None, :] + rotate_half(q) * sin[:, None, :]
        k = k * cos[:, None, :] + rotate_half(k) * sin[:, None, :]
        att = torch.empty_like(q)
        for (s, e) in seg:
            sc = torch.einsum("qhd,khd->hqk", q[s:e], k[s:e]) * (D ** -0.5)
            att[s:e] = torch.einsum("hqk,khd->qhd", torch.softmax(sc, dim=-1), v[s:e])
        x = x + fq(att.reshape(-1, E)) @ sd[p + "attn.proj.weight"].t() + sd[p + "attn.proj.bias"]
        y = fq(F.layer_norm(x, (E,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-6))
        y = y @ sd[p + "mlp.fc1.weight"].t() + sd[p + "mlp.fc1.bias"]
        y = y * torch.sigmoid(1.702 * y)  # quick_gelu
        x = x + fq(y) @ sd[p + "mlp.fc2.weight"].t() + sd[p + "mlp.fc2.bias"]
        if taps is not None and i == 0:
            taps["vit_block0"] = x.clone()
    y = F.layer_norm(x, (E,), sd["visual.merger.ln_q.weight"], sd["visual.merger.ln_q.bias"], 1e-6)
    y = y.reshape(-1, E * cfg.merge * cfg.merge)
    y = F.gelu(y @ sd["visual.merger.mlp.0.weight"].t() + sd["visual.merger.mlp.0.bias"])
    y = y @ sd["visual.merger.mlp.2.weight"].t() + sd["visual.merger.mlp.2.bias"]
    if taps is not None:
        taps["merger"] = y.clone()
    return y


# ----------------------------------------------------------------------------- M-RoPE
def rope_index(cfg: RefConfig, input_ids: Sequence[int], grids) -> Tuple[torch.Tensor, int]:
    """position ids [3, S] for one sequence and the next text position (max + 1)."""
    ids = list(input_ids)
    pos: List[torch.Tensor] = []
    cur, i, g = 0, 0, 0
    n = len(ids)
    while i < n:
        if ids[i] == cfg.image_token_id:
            t, h, w = grids[g]
            g += 1
            lh, lw = h // cfg.merge, w // cfg.merge
            cnt = t * lh * lw
            tt = torch.arange(t).view(-1, 1, 1).expand(t, lh, lw).flatten()
            hh = torch.arange(lh).view(1, -1, 1).expand(t, lh, lw).flatten()
            ww = torch.arange(lw).view(1, 1, -1).expand(t, lh, lw).flatten()
            pos.append(torch.stack([tt, hh, ww]) + cur)
            cur += max(h, w) // cfg.merge
            i += cnt
        else:
            j = i
            while j < n and ids[j] != cfg.image_token_id:
                j += 1
            pos.append(torch.arange(j - i).view(1, -1).expand(3, -1) + cur)
            cur += j - i
            i = j
    p = torch.cat(pos, dim=1)
    return p, int(p.max()) + 1


def mrope_cos_sin(cfg: RefConfig, pos3: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """pos3 [3, S] -> per-token cos/sin rows [S, head_dim] with the (t,h,w) sections selected."""
    D = cfg.head_dim
    inv_freq = 1.0 / (cfg.rope_theta ** (torch.arange(0, D, 2, dtype=torch.float32) / D))
    freqs = pos3.float().unsqueeze(-1) * inv_freq  # [3, S, D/2]
    emb = torch.cat((freqs, freqs), dim=-1)        # [3, S, D]
    cos3, sin3 = emb.cos(), emb.sin()
    sect = list(cfg.mrope_section) * 2
    cos = torch.cat([m[i % 3] for i, m in enumerate(cos3.split(sect, dim=-1))], dim=-1)
    sin = torch.cat([m[i % 3] for i, m in enumerate(sin3.split(sect, dim=-1))], dim=-1)
    return cos, sin


# ----------------------------------------------------------------------------- text decoder
class KVCache:
    def __init__(self, layers: int):
        self.k: List[Optional[torch.Tensor]] = [None] * layers
        self.v: List[Optional[torch.Tensor]] = [None] * layers

    def append(self, i: int, k: torch.Tensor, v: torch.Tensor):
        self.k[i] = k if self.k[i] is None else torch.cat((self.k[i], k), dim=1)
        self.v[i] = v if self.v[i] is None else torch.cat((self.v[i], v), dim=1)
        return self.k[i], self.v[i]


def fake_quant_rows_e4m3(x: torch.Tensor, round_bf16: bool = True) -> torch.Tensor:
    """Per-row dynamic e4m3 quantise -> de-quantise (scale = amax/448, multiply by the reciprocal, RNE) - the
    activation side of the fp8 configuration (product: vis_quant_rows_fp8).  The product quantises bf16 tensors."""
    if round_bf16:
        x = x.to(torch.bfloat16).float()
    sc = (x.abs().amax(dim=-1, keepdim=True) / 448.0).clamp_min(1e-12)
    return (x * (1.0 / sc)).to(torch.float8_e4m3fn).float() * sc


def text_forward(cfg: RefConfig, sd: Dict[str, torch.Tensor], x: torch.Tensor, cos: torch.Tensor,
                 sin: torch.Tensor, cache: KVCache, taps: Optional[dict] = None,
                 n_layers: Optional[int] = None, act_fp8: bool = False) -> torch.Tensor:
    """x [S, hidden] new-token embeddings; cos/sin [S, head_dim]; returns final-normed hidden [S, hidden].
    act_fp8: fake-quantise the input of every projection per row to e4m3 (fp8 prefill configuration; the weights
    in ``sd`` are then expected to be the de-quantised e4m3 weights)."""
    Hq, Hkv, D = cfg.heads, cfg.kv_heads, cfg.head_dim
    S = x.shape[0]
    fq = fake_quant_rows_e4m3 if act_fp8 else (lambda t: t)
    for i in range(cfg.layers if n_layers is None else n_layers):
        p = f"model.layers.{i}."
        y = fq(rms_norm(x, sd[p + "input_layernorm.weight"], cfg.rms_eps))
        q = (y @ sd[p + "self_attn.q_proj.weight"].t() + sd[p + "self_attn.q_proj.bias"]).reshape(S, Hq, D)
        k = (y @ sd[p + "self_attn.k_proj.weight"].t() + sd[p + "self_attn.k_proj.bias"]).reshape(S, Hkv, D)
        v = (y @ sd[p + "self_attn.v_proj.weight"].t() + sd[p + "self_attn.v_proj.bias"]).reshape(S, Hkv, D)
        q = (q * cos[:, None, :] + rotate_half(q) * sin[:, None, :]).transpose(0, 1)  # [Hq,S,D]
        k = (k * cos[:, None, :] + rotate_half(k) * sin[:, None, :]).transpose(0, 1)
        kk, vv = cache.append(i, k, v.transpose(0, 1))
        T = kk.shape[1]
        kr = kk.repeat_interleave(Hq // Hkv, dim=0)
        vr = vv.repeat_interleave(Hq // Hkv, dim=0)
        sc = torch.einsum("hqd,hkd->hqk", q, kr) * (D ** -0.5)
        mask = torch.ones(S, T, dtype=torch.bool).triu(T - S + 1)
        sc = sc.masked_fill(mask, float("-inf"))
        att = torch.einsum("hqk,hkd->qhd", torch.softmax(sc, dim=-1), vr).reshape(S, Hq * D)
        x = x + fq(att) @ sd[p + "self_attn.o_proj.weight"].t()
        y = fq(rms_norm(x, sd[p + "post_attention_layernorm.weight"], cfg.rms_eps))
        g = y @ sd[p + "mlp.gate_proj.weight"].t()
        u = y @ sd[p + "mlp.up_proj.weight"].t()
        x = x + fq(F.silu(g) * u) @ sd[p + "mlp.down_proj.weight"].t()
        if taps is not None and i == 0:
            taps["layer0"] = x.clone()
    return rms_norm(x, sd["model.norm.weight"], cfg.rms_eps)


def embed_inputs(cfg: RefConfig, sd, input_ids: Sequence[int], image_embeds: Optional[torch.Tensor]) -> torch.Tensor:
    ids = torch.tensor(list(input_ids), dtype=torch.long)
    x = sd["model.embed_tokens.weight"][ids].clone()
    if image_embeds is not None:
        m = ids == cfg.image_token_id
        if int(m.sum()) != image_embeds.shape[0]:
            raise ValueError("image tokens and image features do not match")
        x[m] = image_embeds
    return x


def generate(cfg: RefConfig, sd: Dict[str, torch.Tensor], input_ids: Sequence[int],
             pixel_values: Optional[torch.Tensor], grids, max_new_tokens: int,
             eos_ids: Sequence[int] = (), taps: Optional[dict] = None,
             decode_sd: Optional[Dict[str, torch.Tensor]] = None, prefill_fp8_sd: Optional[Dict[str, torch.Tensor]] = None):
    """Greedy decode.  Returns (tokens, per-step logits list [vocab] - entry t produced token t).
    ``decode_sd``: a second state dict used for the per-token steps only (the prompt still runs on ``sd``) - how the
    fp8-decode-weights configuration is checked: decode_sd holds the de-quantised e4m3 projections."""
    if pixel_values is None:
        img = None
    elif prefill_fp8_sd is not None:
        img = vision_forward(cfg, prefill_fp8_sd, pixel_values, grids, taps, act_fp8=True)
    else:
        img = vision_forward(cfg, sd, pixel_values, grids, taps)
    x = embed_inputs(cfg, sd, input_ids, img)
    pos3, next_pos = rope_index(cfg, input_ids, grids or [])
    cos, sin = mrope_cos_sin(cfg, pos3)
    if taps is not None:
        taps["position_ids"] = pos3.clone()
        taps["cos"] = cos.clone()
    cache = KVCache(cfg.layers)
    if prefill_fp8_sd is not None:   # fp8 prefill configuration: de-quantised weights + fake-quantised activations
        h = text_forward(cfg, prefill_fp8_sd, x, cos, sin, cache, taps, act_fp8=True)
    else:
        h = text_forward(cfg, sd, x, cos, sin, cache, taps)
    if taps is not None:
        taps["final_norm_last"] = h[-1].clone()
    logits = h[-1] @ sd["lm_head.weight"].t()
    out: List[int] = []
    all_logits: List[torch.Tensor] = []
    for t in range(max_new_tokens):
        all_logits.append(logits.clone())
        tok = int(torch.argmax(logits))
        out.append(tok)
        if tok in eos_ids or t + 1 == max_new_tokens:
            break
        p = torch.full((3, 1), next_pos + t, dtype=torch.long)
        c, s = mrope_cos_sin(cfg, p)
        dsd = decode_sd if decode_sd is not None else sd
        h = text_forward(cfg, dsd, dsd["model.embed_tokens.weight"][tok][None, :], c, s, cache)
        logits = h[-1] @ dsd["lm_head.weight"].t()
    return out, all_logits
