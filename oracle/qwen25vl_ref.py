"""ORACLE (test infrastructure - never imported by the product path).

CPU fp32 restatement of the Qwen2.5-VL VISION TOWER as shipped in transformers 5.15.0 (``TF:`` =
transformers/models/qwen2_5_vl/modeling_qwen2_5_vl.py) - the model the reference's code default names for both agents
(``Qwen/Qwen2.5-VL-7B-Instruct``: /root/reference utils/config.py:42-45,:59-64; config/models.yaml:6) and the
"windowed attention over image tokens" of the north star:

  window index / window cu_seqlens     TF vision_utils.py get_vision_window_index (:130-188)
  reorder, rope in window order        TF:408-446
  block: RMSNorm, windowed / full attn TF:294-323 (blocks in fullatt_block_indexes attend over the whole image :449-454)
  SwiGLU MLP with biases               TF:85-97
  merger: RMSNorm ln_q + GELU MLP      TF:137-150, then the reverse permutation :464-466

The text decoder, the M-RoPE index (still images: TF:1037-1048 - identical to Qwen2-VL), preprocessing and greedy decoding
are those of oracle/qwen2vl_ref.py, which this file reuses.  Pinned by tests/test_oracle_qwen25.py against vectors
recorded from the real ``Qwen2_5_VLForConditionalGeneration`` (tests/golden/gen_qwen25vl_golden.py ->
qwen25vl_tiny.npz): window index, merged image embeddings, first-step logits, 16 greedy tokens.

Only tests/ may import this.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from . import qwen2vl_ref as R


@dataclass
class Ref25Config(R.RefConfig):
    v_window: int = 112
    v_fullatt: Tuple[int, ...] = (7, 15, 23, 31)


def window_index(cfg: Ref25Config, grids) -> Tuple[torch.Tensor, List[int]]:
    """(permutation of the merge units into window order, cumulative window boundaries in PATCH rows)."""
    win = cfg.v_window // cfg.merge // cfg.patch          # window side in merged tokens
    unit = cfg.merge ** 2
    index_all, cu, base = [], [0], 0
    for (t, h, w) in grids:
        lh, lw = h // cfg.merge, w // cfg.merge
        idx = torch.arange(t * lh * lw).reshape(t, lh, lw)
        ph, pw = win - lh % win, win - lw % win           # (a full extra window of padding when already divisible)
        nh, nw = (lh + ph) // win, (lw + pw) // win
        pad = F.pad(idx, (0, pw, 0, ph), "constant", -100)
        pad = pad.reshape(t, nh, win, nw, win).permute(0, 1, 3, 2, 4).reshape(t, nh * nw, win, win)
        seqlens = (pad != -100).sum([2, 3]).reshape(-1)
        flat = pad.reshape(-1)
        index_all.append(flat[flat != -100] + base)
        for s in (seqlens.cumsum(0) * unit + cu[-1]).tolist():
            cu.append(int(s))
        base += t * lh * lw
    dedup = [cu[0]]
    for c in cu[1:]:
        if c != dedup[-1]:
            dedup.append(c)                                  # empty (all-padding) windows vanish: unique_consecutive
    return torch.cat(index_all), dedup


def vision_forward(cfg: Ref25Config, sd: Dict[str, torch.Tensor], pixel_values: torch.Tensor, grids,
                   taps: Optional[dict] = None) -> torch.Tensor:
    """pixel_values [N, 1176] (merge-block order) -> merged image embeddings [N/4, hidden], original order."""
    E, H, D = cfg.v_embed, cfg.v_heads, cfg.v_head_dim
    unit = cfg.merge ** 2
    x = pixel_values @ sd["visual.patch_embed.proj.weight"].reshape(E, -1).t()
    N = x.shape[0]
    widx, cu_win = window_index(cfg, grids)
    x = x.reshape(N // unit, unit, E)[widx].reshape(N, E)
    cos, sin = R.vision_cos_sin(cfg, grids)
    cos = cos.reshape(N // unit, unit, -1)[widx].reshape(N, -1)
    sin = sin.reshape(N // unit, unit, -1)[widx].reshape(N, -1)
    cu_full, start = [0], 0
    for (t, h, w) in grids:
        for _ in range(t):
            start += h * w
            cu_full.append(start)
    if taps is not None:
        taps["window_index"] = widx.clone()
        taps["cu_window"] = list(cu_win)
    for i in range(cfg.v_depth):
        p = f"visual.blocks.{i}."
        y = R.rms_norm(x, sd[p + "norm1.weight"], 1e-6)
        qkv = (y @ sd[p + "attn.qkv.weight"].t() + sd[p + "attn.qkv.bias"]).reshape(N, 3, H, D)
        q, k, v = qkv[:, 0], qkv[:, 1], qkv[:, 2]
        q = q * cos[:, None, :] + R.rotate_half(q) * sin[:, None, :]
        k = k * cos[:, None, :] + R.rotate_half(k) * sin[:, None, :]
        cu = cu_full if i in cfg.v_fullatt else cu_win
        att = torch.empty_like(q)
        for s, e in zip(cu[:-1], cu[1:]):
            sc = torch.einsum("qhd,khd->hqk", q[s:e], k[s:e]) * (D ** -0.5)
            att[s:e] = torch.einsum("hqk,khd->qhd", torch.softmax(sc, dim=-1), v[s:e])
        x = x + att.reshape(N, E) @ sd[p + "attn.proj.weight"].t() + sd[p + "attn.proj.bias"]
        y = R.rms_norm(x, sd[p + "norm2.weight"], 1e-6)
        g = y @ sd[p + "mlp.gate_proj.weight"].t() + sd[p + "mlp.gate_proj.bias"]
        u = y @ sd[p + "mlp.up_proj.weight"].t() + sd[p + "mlp.up_proj.bias"]
        x = x + (F.silu(g) * u) @ sd[p + "mlp.down_proj.weight"].t() + sd[p + "mlp.down_proj.bias"]
        if taps is not None and i == 0:
            taps["vit_block0"] = x.clone()
    y = R.rms_norm(x, sd["visual.merger.ln_q.weight"], 1e-6).reshape(-1, E * unit)
    y = F.gelu(y @ sd["visual.merger.mlp.0.weight"].t() + sd["visual.merger.mlp.0.bias"])
    y = y @ sd["visual.merger.mlp.2.weight"].t() + sd["visual.merger.mlp.2.bias"]
    y = y[torch.argsort(widx)]
    if taps is not None:
        taps["merger"] = y.clone()
    return y


def generate(cfg: Ref25Config, sd: Dict[str, torch.Tensor], input_ids: Sequence[int], pixel_values: Optional[torch.Tensor],
             grids, max_new_tokens: int, taps: Optional[dict] = None):
    """Greedy decode: this tower + the shared text path of oracle/qwen2vl_ref.py.  Returns (tokens, per-step logits)."""
    img = vision_forward(cfg, sd, pixel_values, grids, taps) if pixel_values is not None else None
    x = R.embed_inputs(cfg, sd, input_ids, img)
    pos3, next_pos = R.rope_index(cfg, input_ids, grids or [])
    cos, sin = R.mrope_cos_sin(cfg, pos3)
    cache = R.KVCache(cfg.layers)
    h = R.text_forward(cfg, sd, x, cos, sin, cache, taps)
    logits = h[-1] @ sd["lm_head.weight"].t()
    out, all_logits = [], []
    for t in range(max_new_tokens):
        all_logits.append(logits.clone())
        tok = int(torch.argmax(logits))
        out.append(tok)
        if t + 1 == max_new_tokens:
            break
        c, s = R.mrope_cos_sin(cfg, torch.full((3, 1), next_pos + t, dtype=torch.long))
        h = R.text_forward(cfg, sd, sd["model.embed_tokens.weight"][tok][None, :], c, s, cache)
        logits = h[-1] @ sd["lm_head.weight"].t()
    return out, all_logits
