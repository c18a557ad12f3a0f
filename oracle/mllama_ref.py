"""ORACLE (test infrastructure - never imported by the product path).

Row f2: CPU restatement, in plain fp32 PyTorch tensor algebra, of the arithmetic the
reference's Auditor endpoint performs when it falls back to ``meta-llama/Llama-3.2-11B-Vision-Instruct``
(reference call sites: src/agents/vlm_auditor.py:81-83 model choice, :152-158 request, :166-234 prompt).
The reference contains no model arithmetic, so the algorithm restated here is the published "mllama" definition
as shipped in transformers 5.15.0 (``TF:`` = transformers/models/mllama/):

  tile canvas / resize / pad / tiles      TF:image_processing_pil_mllama.py:(get_optimal_tiled_canvas,
                                          get_image_size_fit_to_canvas, pad, split_to_tiles_np, pack_images)
  patch embed, CLS, tile + gated pos emb  TF:modeling_mllama.py:102-160,:869-920
  vision layer (LayerNorm, MHA, GELU MLP) TF:modeling_mllama.py:163-313
  aspect-ratio attention mask             TF:modeling_mllama.py:75-99   (only pad x pad pairs are masked)
  global (tanh-gated) layers, concat      TF:modeling_mllama.py:940-1003
  projector                               TF:modeling_mllama.py:1281-1285,:1353-1357
  cross-attention mask / full-row mask    TF:modeling_mllama.py:47-72, TF:processing_mllama.py:34-80
  text self-attn layer, llama3 rope       TF:modeling_mllama.py:469-652,:706-758, TF:modeling_rope_utils.py (llama3)
  cross-attn layer (q/k RMSNorm, gates)   TF:modeling_mllama.py:384-466,:655-703

Scope: one image per prompt (what the reference sends: one text part + one image part per request).
Pinning: tests/test_oracle_mllama.py checks this file against golden vectors produced by the real transformers
modules (tests/golden/gen_mllama_golden.py -> tests/golden/mllama_tiny.npz): preprocessing, cross-attention
states, first-step logits and greedy tokens of a tiny seeded config, for a 2-tile and a 4-tile image, with the
image before and after the prompt text.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


@dataclass
class MllamaRefConfig:
    # text
    hidden: int = 4096
    layers: int = 40
    heads: int = 32
    kv_heads: int = 8
    intermediate: int = 14336
    vocab: int = 128256
    rms_eps: float = 1e-5
    rope_theta: float = 500000.0
    rope_factor: float = 8.0
    rope_low_freq: float = 1.0
    rope_high_freq: float = 4.0
    rope_orig_ctx: int = 8192
    cross_layers: Tuple[int, ...] = (3, 8, 13, 18, 23, 28, 33, 38)
    image_token_id: int = 128256
    # vision
    v_hidden: int = 1280
    v_heads: int = 16
    v_layers: int = 32
    v_global_layers: int = 8
    v_mlp: int = 5120
    v_inter: Tuple[int, ...] = (3, 7, 15, 23, 30)
    v_eps: float = 1e-5
    image_size: int = 560
    patch: int = 14
    max_tiles: int = 4

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads

    @property
    def tile_tokens(self) -> int:          # patches + CLS
        return (self.image_size // self.patch) ** 2 + 1

    @property
    def v_out(self) -> int:
        return self.v_hidden * (1 + len(self.v_inter))


# ----------------------------------------------------------------------------- preprocessing
def supported_aspect_ratios(max_tiles: int) -> List[Tuple[int, int]]:
    return [(w, h) for w in range(1, max_tiles + 1) for h in range(1, max_tiles + 1) if w * h <= max_tiles]


def optimal_canvas(h: int, w: int, max_tiles: int, tile: int) -> Tuple[int, int]:
    """Canvas (height, width): smallest upscale >= 1 if any, else largest downscale; ties -> smallest area."""
    sizes = np.array(supported_aspect_ratios(max_tiles)) * tile
    th, tw = sizes.T
    sh, sw = th / h, tw / w
    scales = np.where(sw > sh, sh, sw)
    up = scales[scales >= 1]
    sel = np.min(up) if len(up) > 0 else np.max(scales[scales < 1])
    chosen = sizes[scales == sel]
    if len(chosen) > 1:
        chosen = chosen[np.argmin(chosen[:, 0] * chosen[:, 1])][None]
    return int(chosen[0][0]), int(chosen[0][1])


def fit_to_canvas(h: int, w: int, ch: int, cw: int, tile: int) -> Tuple[int, int]:
    tw = int(np.clip(w, tile, cw))
    th = int(np.clip(h, tile, ch))
    sh, sw = th / h, tw / w
    if sw < sh:
        return min(math.floor(h * sw) or 1, th), tw
    return th, min(math.floor(w * sh) or 1, tw)


def preprocess_u8(img_hwc_u8: np.ndarray, tile: int, max_tiles: int):
    """uint8 RGB [H,W,3] -> (pixel tiles f32 [max_tiles,3,tile,tile] (absent tiles exact zeros), n_tiles,
    (tiles_h, tiles_w), aspect_ratio_id)."""
    h, w, _ = img_hwc_u8.shape
    ch, cw = optimal_canvas(h, w, max_tiles, tile)
    nth, ntw = ch // tile, cw // tile
    nh, nw = fit_to_canvas(h, w, ch, cw, tile)
    if (nh, nw) != (h, w):
        img_hwc_u8 = np.array(Image.fromarray(img_hwc_u8).resize((nw, nh), resample=Image.Resampling.BILINEAR))
    canvas = np.zeros((ch, cw, 3), dtype=np.float32)          # zero padding happens BEFORE rescale / normalise
    canvas[:nh, :nw] = img_hwc_u8.astype(np.float32)
    x = canvas * np.float32(1.0 / 255.0)
    x = (x - np.array(CLIP_MEAN, dtype=np.float32)) / np.array(CLIP_STD, dtype=np.float32)
    x = x.transpose(2, 0, 1)                                   # [3, ch, cw]
    t = x.reshape(3, nth, tile, ntw, tile).transpose(1, 3, 0, 2, 4).reshape(nth * ntw, 3, tile, tile)
    out = np.zeros((max_tiles, 3, tile, tile), dtype=np.float32)
    out[:nth * ntw] = t
    ar_id = supported_aspect_ratios(max_tiles).index((nth, ntw)) + 1
    return out, nth * ntw, (nth, ntw), ar_id


# ----------------------------------------------------------------------------- small pieces
def rms_norm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    x = x.float()
    return w * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps))


def rotate_half(x: torch.Tensor) -> torch.Tensor:
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def llama3_inv_freq(cfg: MllamaRefConfig) -> torch.Tensor:
    dim = cfg.head_dim
    inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, dim, 2, dtype=torch.int64).float() / dim))
    if cfg.rope_factor is None or cfg.rope_factor <= 0:
        return inv
    low_wl = cfg.rope_orig_ctx / cfg.rope_low_freq
    high_wl = cfg.rope_orig_ctx / cfg.rope_high_freq
    wl = 2 * math.pi / inv
    inv_l = torch.where(wl > low_wl, inv / cfg.rope_factor, inv)
    smooth = (cfg.rope_orig_ctx / wl - cfg.rope_low_freq) / (cfg.rope_high_freq - cfg.rope_low_freq)
    smoothed = (1 - smooth) * inv_l / cfg.rope_factor + smooth * inv_l
    medium = ~(wl < high_wl) * ~(wl > low_wl)
    return torch.where(medium, smoothed, inv_l)


def rope_cos_sin(cfg: MllamaRefConfig, positions: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """positions [S] -> cos, sin [S, head_dim] (f32)."""
    freqs = positions.float()[:, None] * llama3_inv_freq(cfg)[None, :]
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos(), emb.sin()


def _attn(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, mask: Optional[torch.Tensor], scale: float):
    """q [H,Sq,D], k/v [H,Sk,D], mask additive [Sq,Sk] or None -> [Sq, H*D]."""
    s = torch.matmul(q, k.transpose(1, 2)) * scale
    if mask is not None:
        s = s + mask
    p = F.softmax(s, dim=-1, dtype=torch.float32)
    return torch.matmul(p, v).transpose(0, 1).reshape(q.shape[1], -1)


# ----------------------------------------------------------------------------- vision tower
def _vision_layer(sd, pre: str, x: torch.Tensor, mask: torch.Tensor, heads: int, eps: float, gated: bool):
    D = x.shape[-1] // heads
    h = F.layer_norm(x, (x.shape[-1],), sd[pre + "input_layernorm.weight"], sd[pre + "input_layernorm.bias"], eps)
    q = (h @ sd[pre + "self_attn.q_proj.weight"].t()).view(-1, heads, D).transpose(0, 1)
    k = (h @ sd[pre + "self_attn.k_proj.weight"].t()).view(-1, heads, D).transpose(0, 1)
    v = (h @ sd[pre + "self_attn.v_proj.weight"].t()).view(-1, heads, D).transpose(0, 1)
    a = _attn(q, k, v, mask, D ** -0.5) @ sd[pre + "self_attn.o_proj.weight"].t()
    if gated:
        a = sd[pre + "gate_attn"].tanh() * a
    x = x + a
    h = F.layer_norm(x, (x.shape[-1],), sd[pre + "post_attention_layernorm.weight"],
                     sd[pre + "post_attention_layernorm.bias"], eps)
    m = F.gelu(h @ sd[pre + "mlp.fc1.weight"].t() + sd[pre + "mlp.fc1.bias"])
    m = m @ sd[pre + "mlp.fc2.weight"].t() + sd[pre + "mlp.fc2.bias"]
    if gated:
        m = sd[pre + "gate_ffn"].tanh() * m
    return x + m


def vision_forward(cfg: MllamaRefConfig, sd: Dict[str, torch.Tensor], tiles: torch.Tensor, n_tiles: int,
                   ar_id: int, taps: Optional[dict] = None) -> torch.Tensor:
    """tiles f32 [max_tiles,3,S,S] -> cross-attention states [max_tiles * tile_tokens, hidden] (projector applied)."""
    V = "model.vision_model."
    T, E, P = cfg.max_tiles, cfg.v_hidden, cfg.tile_tokens
    w = sd[V + "patch_embedding.weight"]
    x = F.conv2d(tiles, w, stride=cfg.patch).flatten(2).transpose(1, 2)            # [T, P-1, E]
    pre = sd[V + "pre_tile_positional_embedding.embedding.weight"][ar_id].view(T, 1, E)
    x = x + pre * sd[V + "pre_tile_positional_embedding.gate"].tanh()
    x = torch.cat([sd[V + "class_embedding"].view(1, 1, E).expand(T, 1, E), x], dim=1)    # [T, P, E]
    g = sd[V + "gated_positional_embedding.gate"].tanh()
    x = x + (1 - g) * sd[V + "gated_positional_embedding.embedding"].view(1, P, E)
    x = x + g * sd[V + "gated_positional_embedding.tile_embedding.weight"][ar_id].view(T, P, E)
    x = F.layer_norm(x, (E,), sd[V + "layernorm_pre.weight"], sd[V + "layernorm_pre.bias"], 1e-5)
    npad = (8 - P % 8) % 8
    x = F.pad(x, (0, 0, 0, npad))                                                   # [T, P+npad, E]
    L = P + npad
    # mask: 1 marks padding (pad rows of every tile, whole absent tiles); only pad x pad pairs are masked out
    padflag = torch.zeros(T, L)
    padflag[n_tiles:] = 1
    padflag[:, P:] = 1
    padflag = padflag.reshape(T * L, 1)
    mask = padflag @ padflag.t() * torch.finfo(torch.float32).min
    x = x.reshape(T * L, E)
    inter = []
    for i in range(cfg.v_layers):
        x = _vision_layer(sd, f"{V}transformer.layers.{i}.", x, mask, cfg.v_heads, cfg.v_eps, False)
        if i in cfg.v_inter:
            inter.append(x)
    x = F.layer_norm(x, (E,), sd[V + "layernorm_post.weight"], sd[V + "layernorm_post.bias"], 1e-5)
    post = sd[V + "post_tile_positional_embedding.embedding.weight"][ar_id].view(T, 1, E)
    x = (x.view(T, L, E) + post * sd[V + "post_tile_positional_embedding.gate"].tanh()).reshape(T * L, E)
    for i in range(cfg.v_global_layers):
        x = _vision_layer(sd, f"{V}global_transformer.layers.{i}.", x, mask, cfg.v_heads, cfg.v_eps, True)
    x = x.view(T, L, E)[:, :P]
    inter_t = torch.stack(inter, dim=-1).view(T, L, E * len(inter))[:, :P]        # feature index = e * n_inter + i
    feats = torch.cat([x, inter_t], dim=-1).reshape(T * P, cfg.v_out)
    if taps is not None:
        taps["vision_features"] = feats
    return feats @ sd["model.multi_modal_projector.weight"].t() + sd["model.multi_modal_projector.bias"]


# ----------------------------------------------------------------------------- text model
class TextCache:
    def __init__(self, cfg: MllamaRefConfig):
        self.k: Dict[int, torch.Tensor] = {}
        self.v: Dict[int, torch.Tensor] = {}
        self.seen = 0


def text_forward(cfg: MllamaRefConfig, sd: Dict[str, torch.Tensor], x: torch.Tensor, pos0: int,
                 cache: TextCache, cross_states: Optional[torch.Tensor], n_tiles: int, n_masked_rows: int,
                 taps: Optional[dict] = None) -> torch.Tensor:
    """x [S, hidden] (embeddings of tokens at positions pos0..pos0+S-1) -> final-norm hidden states [S, hidden].
    cross_states: projector output [max_tiles*tile_tokens, hidden] on the first call (cached afterwards).
    Rows with absolute position < n_masked_rows precede the image token: their cross-attention sees ALL tiles
    unmasked and their cross-layer MLP contribution is zeroed (the HF full_text_row_masked_out_mask)."""
    L = "model.language_model."
    S, H, KV, D = x.shape[0], cfg.heads, cfg.kv_heads, cfg.head_dim
    G = H // KV
    cos, sin = rope_cos_sin(cfg, torch.arange(pos0, pos0 + S))
    rows = torch.arange(pos0, pos0 + S)
    row_ok = (rows >= n_masked_rows).float()[:, None]
    P = cfg.tile_tokens
    for i in range(cfg.layers):
        pre = f"{L}layers.{i}."
        if i in cfg.cross_layers:
            if cross_states is not None:
                kx = (cross_states @ sd[pre + "cross_attn.k_proj.weight"].t()).view(-1, KV, D).transpose(0, 1)
                vx = (cross_states @ sd[pre + "cross_attn.v_proj.weight"].t()).view(-1, KV, D).transpose(0, 1)
                cache.k[i] = rms_norm(kx, sd[pre + "cross_attn.k_norm.weight"], cfg.rms_eps)
                cache.v[i] = vx
            elif i not in cache.k:
                continue                                  # text-only prompt: cross layers are skipped
            h = rms_norm(x, sd[pre + "input_layernorm.weight"], cfg.rms_eps)
            q = (h @ sd[pre + "cross_attn.q_proj.weight"].t()).view(S, H, D).transpose(0, 1)
            q = rms_norm(q, sd[pre + "cross_attn.q_norm.weight"], cfg.rms_eps)
            kk = cache.k[i].repeat_interleave(G, dim=0)
            vv = cache.v[i].repeat_interleave(G, dim=0)
            mask = torch.zeros(S, kk.shape[1])
            mask[:, n_tiles * P:] = torch.finfo(torch.float32).min    # absent tiles ...
            mask = mask * row_ok                                       # ... except for fully masked rows
            a = _attn(q, kk, vv, mask, D ** -0.5) @ sd[pre + "cross_attn.o_proj.weight"].t()
            x = x + sd[pre + "cross_attn_attn_gate"].tanh() * a
            h = rms_norm(x, sd[pre + "post_attention_layernorm.weight"], cfg.rms_eps)
            m = (F.silu(h @ sd[pre + "mlp.gate_proj.weight"].t()) * (h @ sd[pre + "mlp.up_proj.weight"].t())) \
                @ sd[pre + "mlp.down_proj.weight"].t()
            x = x + sd[pre + "cross_attn_mlp_gate"].tanh() * (row_ok * m)
        else:
            h = rms_norm(x, sd[pre + "input_layernorm.weight"], cfg.rms_eps)
            q = (h @ sd[pre + "self_attn.q_proj.weight"].t()).view(S, H, D).transpose(0, 1)
            k = (h @ sd[pre + "self_attn.k_proj.weight"].t()).view(S, KV, D).transpose(0, 1)
            v = (h @ sd[pre + "self_attn.v_proj.weight"].t()).view(S, KV, D).transpose(0, 1)
            q = q * cos + rotate_half(q) * sin
            k = k * cos + rotate_half(k) * sin
            if i in cache.k:
                k = torch.cat([cache.k[i], k], dim=1)
                v = torch.cat([cache.v[i], v], dim=1)
            cache.k[i], cache.v[i] = k, v
            T = k.shape[1]
            causal = torch.full((S, T), torch.finfo(torch.float32).min).triu(T - S + 1)
            a = _attn(q, k.repeat_interleave(G, dim=0), v.repeat_interleave(G, dim=0), causal, D ** -0.5)
            x = x + a @ sd[pre + "self_attn.o_proj.weight"].t()
            h = rms_norm(x, sd[pre + "post_attention_layernorm.weight"], cfg.rms_eps)
            x = x + (F.silu(h @ sd[pre + "mlp.gate_proj.weight"].t()) * (h @ sd[pre + "mlp.up_proj.weight"].t())) \
                @ sd[pre + "mlp.down_proj.weight"].t()
        if taps is not None and pos0 == 0:
            taps[f"layer{i}"] = x
    cache.seen = pos0 + S
    return rms_norm(x, sd[L + "norm.weight"], cfg.rms_eps)


def masked_rows(cfg: MllamaRefConfig, input_ids: Sequence[int]) -> int:
    """Number of leading text rows that precede the (single) image token; S when there is no image."""
    locs = [i for i, t in enumerate(input_ids) if t == cfg.image_token_id]
    if len(locs) > 1:
        raise ValueError("oracle scope: one image per prompt")
    return locs[0] if locs else len(input_ids)


def generate(cfg: MllamaRefConfig, sd: Dict[str, torch.Tensor], input_ids: Sequence[int],
             image_u8: Optional[np.ndarray], max_new_tokens: int, eos_ids: Sequence[int] = (),
             taps: Optional[dict] = None) -> Tuple[List[int], List[torch.Tensor]]:
    """Greedy generation.  Returns (new tokens, per-step f32 logits)."""
    with torch.no_grad():
        cross, n_tiles = None, 0
        if image_u8 is not None:
            tiles, n_tiles, _, ar_id = preprocess_u8(image_u8, cfg.image_size, cfg.max_tiles)
            cross = vision_forward(cfg, sd, torch.from_numpy(tiles), n_tiles, ar_id, taps)
            if taps is not None:
                taps["cross_states"] = cross
        nm = masked_rows(cfg, input_ids) if image_u8 is not None else 0
        emb = sd["model.language_model.embed_tokens.weight"]
        cache = TextCache(cfg)
        x = emb[torch.tensor(list(input_ids))]
        h = text_forward(cfg, sd, x, 0, cache, cross, n_tiles, nm, taps)
        out, logits_all = [], []
        logits = h[-1] @ sd["lm_head.weight"].t()
        for _ in range(max_new_tokens):
            logits_all.append(logits)
            tok = int(torch.argmax(logits))
            out.append(tok)
            if tok in eos_ids or len(out) == max_new_tokens:
                break
            h = text_forward(cfg, sd, emb[torch.tensor([tok])], cache.seen, cache, None, n_tiles, nm)
            logits = h[-1] @ sd["lm_head.weight"].t()
        return out, logits_all
