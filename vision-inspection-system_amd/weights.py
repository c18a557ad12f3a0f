"""Weights: naming, synthetic generation, safetensors loading and the HBM layouts the kernels read.

Checkpoint tensor names follow the published Qwen2-VL checkpoints (``visual.*``, ``model.*``,
``lm_head.weight``; the newer ``model.visual.*`` / ``model.language_model.*`` prefixes are accepted).

Layout decisions (DESIGN.md "Data layout in HBM"):
  * every nn.Linear weight stays [out, in] (K-contiguous) in bf16 - the "NT" GEMM operand;
  * q/k/v projections are concatenated into one [Hq*D + 2*Hkv*D, hidden] matrix (one GEMM);
  * gate/up projections are interleaved in 16-row groups ([gate 16 | up 16] ...) so the
    SwiGLU product happens in the GEMM/GEMV epilogue (same lane holds gate_i and up_i);
  * the patch-embed Conv3d weight [E, 3, 2, 14, 14] is flattened to [E, 1176] and zero-padded
    to 1216 columns (a multiple of the GEMM K-step of 64).
"""
from __future__ import annotations

import glob
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np
import torch

from .config import Qwen2VLConfig

PATCH_K_PAD = 64  # GEMM K-step


def interleave_gate_up(gate: torch.Tensor, up: torch.Tensor) -> torch.Tensor:
    """[I,K],[I,K] -> [2I,K] with rows [g0..g15, u0..u15, g16..g31, u16..u31, ...]. I % 16 == 0."""
    I, K = gate.shape
    if up.shape != gate.shape or I % 16 != 0:
        raise ValueError("gate/up must have equal shapes with rows % 16 == 0")
    g = gate.reshape(I // 16, 16, K)
    u = up.reshape(I // 16, 16, K)
    return torch.stack((g, u), dim=1).reshape(2 * I, K).contiguous()


def pad_cols(w: torch.Tensor, multiple: int) -> torch.Tensor:
    """Zero-pad the last dimension of a 2-D tensor up to a multiple."""
    n, k = w.shape
    kp = (k + multiple - 1) // multiple * multiple
    if kp == k:
        return w.contiguous()
    out = torch.zeros((n, kp), dtype=w.dtype, device=w.device)
    out[:, :k] = w
    return out


# ----------------------------------------------------------------------------- names / shapes
def tensor_shapes(cfg: Qwen2VLConfig) -> Dict[str, tuple]:
    """Checkpoint tensor name -> shape, in checkpoint (HF) naming."""
    E, H = cfg.v_embed, cfg.hidden
    D = cfg.head_dim
    s: Dict[str, tuple] = {"visual.patch_embed.proj.weight": (E, 3, cfg.temporal, cfg.patch, cfg.patch)}
    v25 = cfg.vision_arch == "qwen2_5_vl"
    for i in range(cfg.v_depth):
        p = f"visual.blocks.{i}."
        s[p + "norm1.weight"] = (E,)
        s[p + "norm2.weight"] = (E,)
        s[p + "attn.qkv.weight"] = (3 * E, E); s[p + "attn.qkv.bias"] = (3 * E,)
        s[p + "attn.proj.weight"] = (E, E); s[p + "attn.proj.bias"] = (E,)
        if v25:     # RMSNorm (no bias), SwiGLU MLP with biases (TF:models/qwen2_5_vl/modeling_qwen2_5_vl.py:85-97,:294-323)
            for nm in ("gate_proj", "up_proj"):
                s[p + f"mlp.{nm}.weight"] = (cfg.v_mlp, E); s[p + f"mlp.{nm}.bias"] = (cfg.v_mlp,)
            s[p + "mlp.down_proj.weight"] = (E, cfg.v_mlp); s[p + "mlp.down_proj.bias"] = (E,)
        else:
            s[p + "norm1.bias"] = (E,); s[p + "norm2.bias"] = (E,)
            s[p + "mlp.fc1.weight"] = (cfg.v_mlp, E); s[p + "mlp.fc1.bias"] = (cfg.v_mlp,)
            s[p + "mlp.fc2.weight"] = (E, cfg.v_mlp); s[p + "mlp.fc2.bias"] = (E,)
    M = E * cfg.merge ** 2
    s["visual.merger.ln_q.weight"] = (E,)
    if not v25:
        s["visual.merger.ln_q.bias"] = (E,)
    s["visual.merger.mlp.0.weight"] = (M, M); s["visual.merger.mlp.0.bias"] = (M,)
    s["visual.merger.mlp.2.weight"] = (H, M); s["visual.merger.mlp.2.bias"] = (H,)
    s["model.embed_tokens.weight"] = (cfg.vocab, H)
    for i in range(cfg.layers):
        p = f"model.layers.{i}."
        s[p + "input_layernorm.weight"] = (H,)
        s[p + "post_attention_layernorm.weight"] = (H,)
        s[p + "self_attn.q_proj.weight"] = (cfg.heads * D, H); s[p + "self_attn.q_proj.bias"] = (cfg.heads * D,)
        s[p + "self_attn.k_proj.weight"] = (cfg.kv_heads * D, H); s[p + "self_attn.k_proj.bias"] = (cfg.kv_heads * D,)
        s[p + "self_attn.v_proj.weight"] = (cfg.kv_heads * D, H); s[p + "self_attn.v_proj.bias"] = (cfg.kv_heads * D,)
        s[p + "self_attn.o_proj.weight"] = (H, cfg.heads * D)
        s[p + "mlp.gate_proj.weight"] = (cfg.intermediate, H)
        s[p + "mlp.up_proj.weight"] = (cfg.intermediate, H)
        s[p + "mlp.down_proj.weight"] = (H, cfg.intermediate)
    s["model.norm.weight"] = (H,)
    s["lm_head.weight"] = (cfg.vocab, H)
    return s


# ----------------------------------------------------------------------------- synthetic weights
def _hash_uniform(n: int, stream: int) -> np.ndarray:
    """n floats in [-1, 1): counter-based splitmix64 - bit-identical on every machine/library version."""
    with np.errstate(over="ignore"):
        z = np.arange(n, dtype=np.uint64) + np.uint64((stream * 0x9E3779B97F4A7C15 + 0x1234567) % (1 << 64))
        z = (z + np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (2.0 / (1 << 53)) - 1.0


def synth_state_dict(cfg: Qwen2VLConfig, seed: int = 0, rng: str = "hash", device=None) -> Dict[str, torch.Tensor]:
    """Deterministic fp32 CPU weights (variance-preserving scales) in checkpoint naming.
    ``rng="torch"`` draws from a seeded torch generator (on ``device`` when given; the result is always returned on
    the CPU) instead of the portable hash: for the full-size (7B-shape) parity tests, where 10^9 values are needed
    within seconds and nothing is compared across machines.

    Used for the tiny parity model: the same dict feeds the oracle, the transformers golden
    generator and the HIP engine.  Values are rounded to bf16-representable numbers so every
    consumer sees exactly the same weights.
    """
    import zlib
    out: Dict[str, torch.Tensor] = {}
    gen = torch.Generator(device=device or "cpu").manual_seed(seed) if rng == "torch" else None
    for name, shape in tensor_shapes(cfg).items():
        n = int(np.prod(shape))
        if gen is not None:
            u = torch.rand(n, generator=gen, dtype=torch.float32, device=gen.device).mul_(2.0).sub_(1.0)
        else:
            u = _hash_uniform(n, seed * 100003 + (zlib.crc32(name.encode()) & 0xFFFFFFF))
        if name.endswith("norm.weight") or name.endswith("norm1.weight") or name.endswith("norm2.weight") \
                or name.endswith("layernorm.weight") or name.endswith("ln_q.weight"):
            v = 1.0 + 0.1 * u
        elif name.endswith(".bias"):
            v = 0.1 * u
        elif name.endswith("embed_tokens.weight"):
            v = u * float(np.sqrt(3.0))  # unit-variance rows
        else:
            fan_in = int(np.prod(shape[1:]))
            v = u * float(np.sqrt(3.0 / fan_in))
        t = v.reshape(shape) if gen is not None else torch.from_numpy(v.astype(np.float32).reshape(shape))
        out[name] = t.to(torch.bfloat16).float().cpu()
    return out


# ----------------------------------------------------------------------------- device layouts
@dataclass
class VitBlockWeights:
    """Qwen2-VL: LayerNorm weights + biases, fc1 / fc2.  Qwen2.5-VL: ln*_b are None (RMSNorm); fc1_w / fc1_b hold the
    gate / up projections interleaved in 16-row groups (SwiGLU in the GEMM epilogue) and fc2_w the down projection, all
    zero-padded from v_mlp to cfg.v_mlp_pad."""
    ln1_w: torch.Tensor; ln1_b: Optional[torch.Tensor]; ln2_w: torch.Tensor; ln2_b: Optional[torch.Tensor]
    qkv_w: torch.Tensor; qkv_b: torch.Tensor; proj_w: torch.Tensor; proj_b: torch.Tensor
    fc1_w: torch.Tensor; fc1_b: torch.Tensor; fc2_w: torch.Tensor; fc2_b: torch.Tensor


@dataclass
class LlmLayerWeights:
    ln1_w: torch.Tensor; ln2_w: torch.Tensor
    qkv_w: torch.Tensor; qkv_b: torch.Tensor; o_w: torch.Tensor
    gateup_w: torch.Tensor; down_w: torch.Tensor


@dataclass
class DeviceWeights:
    patch_w: torch.Tensor
    vit: List[VitBlockWeights]
    merger_ln_w: torch.Tensor; merger_ln_b: Optional[torch.Tensor]
    merger_fc0_w: torch.Tensor; merger_fc0_b: torch.Tensor
    merger_fc2_w: torch.Tensor; merger_fc2_b: torch.Tensor
    embed: torch.Tensor
    llm: List[LlmLayerWeights]
    final_norm_w: torch.Tensor
    lm_head: torch.Tensor

    def nbytes(self) -> int:
        tot = 0
        def acc(o):
            nonlocal tot
            for v in vars(o).values():
                if isinstance(v, torch.Tensor):
                    tot += v.numel() * v.element_size()
                elif isinstance(v, list):
                    for e in v:
                        acc(e)
        acc(self)
        return tot


def _norm_key(k: str) -> str:
    for old, new in (("model.language_model.", "model."), ("model.visual.", "visual."),
                     ("language_model.model.", "model."), ("language_model.lm_head.", "lm_head.")):
        if k.startswith(old):
            return new + k[len(old):]
    return k


def pack_device_weights(cfg: Qwen2VLConfig, sd: Dict[str, torch.Tensor], device) -> DeviceWeights:
    """Checkpoint-named tensors (any float dtype, any device) -> bf16 kernel layouts on ``device``."""
    sd = {_norm_key(k): v for k, v in sd.items()}
    missing = [k for k in tensor_shapes(cfg) if k not in sd and k != "lm_head.weight"]
    if missing:
        raise KeyError(f"checkpoint is missing {len(missing)} tensors, e.g. {missing[:3]}")

    def dv(name: str) -> torch.Tensor:
        return sd[name].to(device=device, dtype=torch.bfloat16).contiguous()

    E = cfg.v_embed
    patch_w = pad_cols(dv("visual.patch_embed.proj.weight").reshape(E, -1), PATCH_K_PAD)
    vit = []
    v25 = cfg.vision_arch == "qwen2_5_vl"
    pad = cfg.v_mlp_pad

    def pad_rows(t: torch.Tensor) -> torch.Tensor:     # [v_mlp, ...] -> [v_mlp_pad, ...], zero rows appended
        if t.shape[0] == pad:
            return t
        out = torch.zeros((pad,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        out[:t.shape[0]] = t
        return out

    for i in range(cfg.v_depth):
        p = f"visual.blocks.{i}."
        if v25:
            gu_w = interleave_gate_up(pad_rows(dv(p + "mlp.gate_proj.weight")), pad_rows(dv(p + "mlp.up_proj.weight")))
            gu_b = interleave_gate_up(pad_rows(dv(p + "mlp.gate_proj.bias")).view(-1, 1),
                                      pad_rows(dv(p + "mlp.up_proj.bias")).view(-1, 1)).view(-1).contiguous()
            vit.append(VitBlockWeights(
                dv(p + "norm1.weight"), None, dv(p + "norm2.weight"), None,
                dv(p + "attn.qkv.weight"), dv(p + "attn.qkv.bias"), dv(p + "attn.proj.weight"), dv(p + "attn.proj.bias"),
                gu_w, gu_b, pad_cols(dv(p + "mlp.down_proj.weight"), 64), dv(p + "mlp.down_proj.bias")))
            continue
        vit.append(VitBlockWeights(
            dv(p + "norm1.weight"), dv(p + "norm1.bias"), dv(p + "norm2.weight"), dv(p + "norm2.bias"),
            dv(p + "attn.qkv.weight"), dv(p + "attn.qkv.bias"), dv(p + "attn.proj.weight"), dv(p + "attn.proj.bias"),
            dv(p + "mlp.fc1.weight"), dv(p + "mlp.fc1.bias"), dv(p + "mlp.fc2.weight"), dv(p + "mlp.fc2.bias")))
    llm = []
    for i in range(cfg.layers):
        p = f"model.layers.{i}."
        qkv_w = torch.cat([dv(p + "self_attn.q_proj.weight"), dv(p + "self_attn.k_proj.weight"),
                           dv(p + "self_attn.v_proj.weight")], dim=0).contiguous()
        qkv_b = torch.cat([dv(p + "self_attn.q_proj.bias"), dv(p + "self_attn.k_proj.bias"),
                           dv(p + "self_attn.v_proj.bias")], dim=0).contiguous()
        gu = interleave_gate_up(dv(p + "mlp.gate_proj.weight"), dv(p + "mlp.up_proj.weight"))
        llm.append(LlmLayerWeights(dv(p + "input_layernorm.weight"), dv(p + "post_attention_layernorm.weight"),
                                   qkv_w, qkv_b, dv(p + "self_attn.o_proj.weight"), gu,
                                   dv(p + "mlp.down_proj.weight")))
    embed = dv("model.embed_tokens.weight")
    lm_head = dv("lm_head.weight") if "lm_head.weight" in sd else embed  # tied checkpoints
    return DeviceWeights(patch_w, vit, dv("visual.merger.ln_q.weight"), None if v25 else dv("visual.merger.ln_q.bias"),
                         dv("visual.merger.mlp.0.weight"), dv("visual.merger.mlp.0.bias"),
                         dv("visual.merger.mlp.2.weight"), dv("visual.merger.mlp.2.bias"),
                         embed, llm, dv("model.norm.weight"), lm_head)


def random_device_weights(cfg: Qwen2VLConfig, device, seed: int = 0, std: float = 0.02, scaled: bool = False,
                          branch_gain: float = 1.0) -> DeviceWeights:
    """Seeded normal(0, std) bf16 weights generated ON the device at the exact shapes of ``cfg``.

    This is what the throughput benchmark uses (no checkpoint exists offline):
    timing and roofline numbers are valid, generated text is noise.  Norm weights are 1.

    ``scaled=True``: variance-preserving values instead (every matrix normal(0, 1 / fan_in), unit-variance embedding
    rows, norm weights 1 +- 0.1, biases +- 0.1 - the distribution family of ``synth_state_dict``), generated without a CPU
    copy: for the FULL-depth precision tests, where activations and logits must stay O(1) through 28 + 32 layers.
    ``branch_gain`` scales the matrices that write into the residual stream (attention output and MLP down / fc2
    projections) - 1 / sqrt(2 L) is the usual initialisation of trained transformers (GPT-2 style)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)

    def rn(*shape, s=None, gain=1.0):
        if s is None:
            s = std if not scaled else (0.1 if len(shape) == 1 else float(shape[-1]) ** -0.5)
        return (torch.randn(shape, generator=g, device=device, dtype=torch.float32) * (s * gain)).to(torch.bfloat16)

    def ones(n):
        if scaled:
            return (1.0 + 0.1 * (2.0 * torch.rand(n, generator=g, device=device, dtype=torch.float32) - 1.0)).to(torch.bfloat16)
        return torch.ones(n, dtype=torch.bfloat16, device=device)

    def zeros(n):
        return rn(n) if scaled else torch.zeros(n, dtype=torch.bfloat16, device=device)

    bg = branch_gain
    E, H, D = cfg.v_embed, cfg.hidden, cfg.head_dim
    M = E * cfg.merge ** 2
    patch_w = pad_cols(rn(E, cfg.patch_dim), PATCH_K_PAD)
    if cfg.vision_arch == "qwen2_5_vl":
        pad, I = cfg.v_mlp_pad, cfg.v_mlp

        def padded(t, dim):          # zero the pad rows / columns (exactly what pack_device_weights produces)
            if dim == 0:
                t[I:] = 0
            else:
                t[:, I:] = 0
            return t
        vit = []
        for _ in range(cfg.v_depth):
            g_w, u_w = padded(rn(pad, E), 0), padded(rn(pad, E), 0)
            g_b, u_b = padded(rn(pad), 0), padded(rn(pad), 0)
            vit.append(VitBlockWeights(ones(E), None, ones(E), None, rn(3 * E, E), rn(3 * E), rn(E, E, gain=bg), rn(E),
                                       interleave_gate_up(g_w, u_w),
                                       interleave_gate_up(g_b.view(-1, 1), u_b.view(-1, 1)).view(-1).contiguous(),
                                       padded(rn(E, pad, gain=bg), 1), rn(E)))
    else:
        vit = [VitBlockWeights(ones(E), zeros(E), ones(E), zeros(E), rn(3 * E, E), rn(3 * E), rn(E, E, gain=bg), rn(E),
                               rn(cfg.v_mlp, E), rn(cfg.v_mlp), rn(E, cfg.v_mlp, gain=bg), rn(E))
               for _ in range(cfg.v_depth)]
    llm = []
    for _ in range(cfg.layers):
        nq = (cfg.heads + 2 * cfg.kv_heads) * D
        llm.append(LlmLayerWeights(ones(H), ones(H), rn(nq, H), rn(nq), rn(H, cfg.heads * D, gain=bg),
                                   rn(2 * cfg.intermediate, H), rn(H, cfg.intermediate, gain=bg)))
    return DeviceWeights(patch_w, vit, ones(E), None if cfg.vision_arch == "qwen2_5_vl" else zeros(E), rn(M, M), rn(M), rn(H, M), rn(H),
                         rn(cfg.vocab, H, s=1.0 if scaled else None), llm, ones(H), rn(cfg.vocab, H))


def load_safetensors_dir(cfg: Qwen2VLConfig, path: str, device) -> DeviceWeights:
    """Load ``*.safetensors`` shards from a LOCAL model directory (no hub access, ever)."""
    from safetensors import safe_open
    files = sorted(glob.glob(os.path.join(path, "*.safetensors")))
    if not files:
        raise FileNotFoundError(f"no *.safetensors under {path} (the local backend only loads local files)")
    sd: Dict[str, torch.Tensor] = {}
    for fpath in files:
        with safe_open(fpath, framework="pt", device="cpu") as f:
            for k in f.keys():
                sd[k] = f.get_tensor(k)
    return pack_device_weights(cfg, sd, device)
