"""Row f2: the Auditor's fallback model, Llama-3.2-11B-Vision ("mllama"): configuration,
checkpoint tensor names/shapes, deterministic synthetic weights and the HBM layouts the kernels read.

Reference: src/agents/vlm_auditor.py:81-83 (``meta-llama/Llama-3.2-11B-Vision-Instruct`` when Groq is absent).
Tensor names follow the published checkpoint as transformers 5.15 loads it (``model.vision_model.*``,
``model.language_model.*``, ``model.multi_modal_projector.*``, ``lm_head.weight``).

Layout decisions on top of weights.py's (everything [out, in] bf16, fused qkv, 16-row interleaved gate/up):
  * every tanh gate is a scalar known at load time, so it is FOLDED into the projection that feeds the gated
    residual add (vision global layers: o_proj / fc2 (+bias); text cross layers: cross o_proj / down_proj) -
    no kernel knows about gates;
  * the three vision embedding tables (pre-tile, gated position + tile position, post-tile) are combined on the
    host into two per-aspect-ratio tables [max_ar+1][tiles][tile_tokens][E] / [max_ar+1][tiles][E] in f32 -> bf16;
  * cross-attention K/V projections of a layer are one [2*Hkv*D, hidden] matrix (one GEMM over the vision
    tokens), their K/V stay resident per request ("cross cache");
  * the patch-embed Conv2d weight [E, 3, 14, 14] is flattened to [E, 588] and zero-padded to 640 columns.
"""
from __future__ import annotations

import zlib
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from .weights import _hash_uniform, interleave_gate_up, pad_cols


@dataclass
class MllamaConfig:
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
    eos_ids: Tuple[int, ...] = (128001, 128008, 128009)
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
    image_mean: Tuple[float, float, float] = (0.48145466, 0.4578275, 0.40821073)
    image_std: Tuple[float, float, float] = (0.26862954, 0.26130258, 0.27577711)
    name: str = "llama-3.2-11b-vision"

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads

    @property
    def v_head_dim(self) -> int:
        return self.v_hidden // self.v_heads

    @property
    def tile_tokens(self) -> int:
        return (self.image_size // self.patch) ** 2 + 1

    @property
    def v_out(self) -> int:
        return self.v_hidden * (1 + len(self.v_inter))

    @property
    def max_ar_id(self) -> int:
        return len([(w, h) for w in range(1, self.max_tiles + 1) for h in range(1, self.max_tiles + 1)
                    if w * h <= self.max_tiles])

    @classmethod
    def mllama_11b(cls) -> "MllamaConfig":
        return cls()

    @classmethod
    def tiny(cls) -> "MllamaConfig":
        """Kernel-compatible miniature (head dims 128 / 80 like the real model) for parity tests."""
        return cls(hidden=256, layers=5, heads=2, kv_heads=1, intermediate=704, vocab=512, cross_layers=(1, 3),
                   image_token_id=512, eos_ids=(2,), rope_orig_ctx=64, v_hidden=320, v_heads=4, v_layers=4,
                   v_global_layers=2, v_mlp=1280, v_inter=(1, 3), image_size=56, name="mllama-tiny")

    def validate_for_kernels(self) -> None:
        if self.head_dim != 128 or self.v_head_dim != 80:
            raise ValueError("kernels are specialised for head_dim 128 (text) and 80 (vision)")
        if self.hidden % 64 or self.v_hidden % 64 or self.intermediate % 64 or self.v_mlp % 64 or self.v_out % 64:
            raise ValueError("GEMM K dimensions must be multiples of 64")
        if self.intermediate % 16 or self.heads % self.kv_heads or self.heads // self.kv_heads not in (1, 2, 4, 7, 8):
            raise ValueError("unsupported GQA group / intermediate size")


# ----------------------------------------------------------------------------- names / shapes
def tensor_shapes(cfg: MllamaConfig) -> Dict[str, tuple]:
    E, H, D = cfg.v_hidden, cfg.hidden, cfg.head_dim
    T, P = cfg.max_tiles, cfg.tile_tokens
    A = cfg.max_ar_id + 1
    V = "model.vision_model."
    s: Dict[str, tuple] = {
        V + "patch_embedding.weight": (E, 3, cfg.patch, cfg.patch),
        V + "class_embedding": (E,),
        V + "gated_positional_embedding.gate": (1,),
        V + "gated_positional_embedding.embedding": (P, E),
        V + "gated_positional_embedding.tile_embedding.weight": (A, T * P * E),
        V + "pre_tile_positional_embedding.embedding.weight": (A, T * E),
        V + "pre_tile_positional_embedding.gate": (1,),
        V + "post_tile_positional_embedding.embedding.weight": (A, T * E),
        V + "post_tile_positional_embedding.gate": (1,),
        V + "layernorm_pre.weight": (E,), V + "layernorm_pre.bias": (E,),
        V + "layernorm_post.weight": (E,), V + "layernorm_post.bias": (E,),
    }
    for stack, n, gated in (("transformer", cfg.v_layers, False), ("global_transformer", cfg.v_global_layers, True)):
        for i in range(n):
            p = f"{V}{stack}.layers.{i}."
            if gated:
                s[p + "gate_attn"] = (1,); s[p + "gate_ffn"] = (1,)
            for nm in ("q", "k", "v", "o"):
                s[p + f"self_attn.{nm}_proj.weight"] = (E, E)
            s[p + "mlp.fc1.weight"] = (cfg.v_mlp, E); s[p + "mlp.fc1.bias"] = (cfg.v_mlp,)
            s[p + "mlp.fc2.weight"] = (E, cfg.v_mlp); s[p + "mlp.fc2.bias"] = (E,)
            s[p + "input_layernorm.weight"] = (E,); s[p + "input_layernorm.bias"] = (E,)
            s[p + "post_attention_layernorm.weight"] = (E,); s[p + "post_attention_layernorm.bias"] = (E,)
    L = "model.language_model."
    s[L + "embed_tokens.weight"] = (cfg.vocab + 8, H)
    for i in range(cfg.layers):
        p = f"{L}layers.{i}."
        a = "cross_attn" if i in cfg.cross_layers else "self_attn"
        if i in cfg.cross_layers:
            s[p + "cross_attn_attn_gate"] = (1,); s[p + "cross_attn_mlp_gate"] = (1,)
            s[p + "cross_attn.q_norm.weight"] = (D,); s[p + "cross_attn.k_norm.weight"] = (D,)
        s[p + f"{a}.q_proj.weight"] = (cfg.heads * D, H)
        s[p + f"{a}.k_proj.weight"] = (cfg.kv_heads * D, H)
        s[p + f"{a}.v_proj.weight"] = (cfg.kv_heads * D, H)
        s[p + f"{a}.o_proj.weight"] = (H, cfg.heads * D)
        s[p + "mlp.gate_proj.weight"] = (cfg.intermediate, H)
        s[p + "mlp.up_proj.weight"] = (cfg.intermediate, H)
        s[p + "mlp.down_proj.weight"] = (H, cfg.intermediate)
        s[p + "input_layernorm.weight"] = (H,)
        s[p + "post_attention_layernorm.weight"] = (H,)
    s[L + "norm.weight"] = (H,)
    s["model.multi_modal_projector.weight"] = (H, cfg.v_out)
    s["model.multi_modal_projector.bias"] = (H,)
    s["lm_head.weight"] = (cfg.vocab, H)
    return s


def synth_state_dict(cfg: MllamaConfig, seed: int = 0, rng: str = "hash", device=None) -> Dict[str, torch.Tensor]:
    """Deterministic fp32 CPU weights in checkpoint naming, rounded to bf16-representable values (the oracle, the
    transformers golden generator and the HIP engine all consume exactly these numbers).  Gates are non-trivial
    (|gate| up to 0.8) so every gated path is exercised."""
    out: Dict[str, torch.Tensor] = {}
    gen = torch.Generator(device=device or "cpu").manual_seed(seed) if rng == "torch" else None      # fast path for 11B-shape tests
    for name, shape in tensor_shapes(cfg).items():
        n = int(np.prod(shape))
        if gen is not None:
            u = torch.rand(n, generator=gen, dtype=torch.float32, device=gen.device).mul_(2.0).sub_(1.0)
        else:
            u = _hash_uniform(n, seed * 100003 + (zlib.crc32(name.encode()) & 0xFFFFFFF))
        if name.endswith("layernorm.weight") or name.endswith("norm.weight") or name.endswith("layernorm_pre.weight") \
                or name.endswith("layernorm_post.weight"):
            v = 1.0 + 0.1 * u
        elif name.endswith(".bias"):
            v = 0.1 * u
        elif name.endswith("gate") or name.endswith("gate_attn") or name.endswith("gate_ffn"):
            v = 0.3 + 0.5 * u
        elif name.endswith("embed_tokens.weight"):
            v = u * float(np.sqrt(3.0))
        elif "positional_embedding" in name or name.endswith("class_embedding"):
            v = 0.5 * u
        else:
            fan_in = int(np.prod(shape[1:]))
            v = u * float(np.sqrt(3.0 / fan_in))
        t = v.reshape(shape) if gen is not None else torch.from_numpy(v.astype(np.float32).reshape(shape))
        out[name] = t.to(torch.bfloat16).float().cpu()
    return out


# ----------------------------------------------------------------------------- device layouts
@dataclass
class MllamaVisionLayer:
    ln1_w: torch.Tensor; ln1_b: torch.Tensor; ln2_w: torch.Tensor; ln2_b: torch.Tensor
    qkv_w: torch.Tensor; o_w: torch.Tensor
    fc1_w: torch.Tensor; fc1_b: torch.Tensor; fc2_w: torch.Tensor; fc2_b: torch.Tensor


@dataclass
class MllamaTextLayer:
    cross: bool
    ln1_w: torch.Tensor; ln2_w: torch.Tensor
    qkv_w: Optional[torch.Tensor]      # self: [Hq*D + 2*Hkv*D, H]; cross: q_proj only [Hq*D, H]
    kv_w: Optional[torch.Tensor]       # cross: [2*Hkv*D, H] applied to the vision tokens
    q_norm: Optional[torch.Tensor]; k_norm: Optional[torch.Tensor]
    o_w: torch.Tensor                  # cross: pre-multiplied by tanh(attn gate)
    gateup_w: torch.Tensor
    down_w: torch.Tensor               # cross: pre-multiplied by tanh(mlp gate)


@dataclass
class MllamaDeviceWeights:
    patch_w: torch.Tensor                       # [E, 640]
    cls_pos: torch.Tensor                       # [A][T][P][E] bf16: everything added to the patch/CLS rows before ln_pre
    post_tile: torch.Tensor                     # [A][T][E] bf16 (gated)
    ln_pre_w: torch.Tensor; ln_pre_b: torch.Tensor; ln_post_w: torch.Tensor; ln_post_b: torch.Tensor
    v_layers: List[MllamaVisionLayer] = field(default_factory=list)
    v_global: List[MllamaVisionLayer] = field(default_factory=list)
    proj_w: torch.Tensor = None; proj_b: torch.Tensor = None      # projector over the PERMUTED feature order
    embed: torch.Tensor = None
    layers: List[MllamaTextLayer] = field(default_factory=list)
    norm_w: torch.Tensor = None
    lm_head: torch.Tensor = None


def pack_device_weights(cfg: MllamaConfig, sd: Dict[str, torch.Tensor], device) -> MllamaDeviceWeights:
    """fp32 checkpoint-named tensors -> bf16 device tensors in kernel layouts (see module docstring)."""
    bf = torch.bfloat16
    dev = torch.device(device)
    V, L = "model.vision_model.", "model.language_model."
    E, T, P, A = cfg.v_hidden, cfg.max_tiles, cfg.tile_tokens, cfg.max_ar_id + 1

    def d(t):
        return t.to(bf).contiguous().to(dev)

    # embeddings added before layernorm_pre: row 0 of a tile = class embedding, rows 1.. = conv output
    g = torch.tanh(sd[V + "gated_positional_embedding.gate"].float())
    pre = sd[V + "pre_tile_positional_embedding.embedding.weight"].float().view(A, T, 1, E) * \
        torch.tanh(sd[V + "pre_tile_positional_embedding.gate"].float())
    pos = (1 - g) * sd[V + "gated_positional_embedding.embedding"].float().view(1, 1, P, E) + \
        g * sd[V + "gated_positional_embedding.tile_embedding.weight"].float().view(A, T, P, E)
    cls_pos = pos.clone()
    cls_pos[:, :, 1:] += pre                                   # patches get the pre-tile embedding, CLS does not
    cls_pos[:, :, 0] += sd[V + "class_embedding"].float().view(1, 1, E)
    post = sd[V + "post_tile_positional_embedding.embedding.weight"].float().view(A, T, E) * \
        torch.tanh(sd[V + "post_tile_positional_embedding.gate"].float())

    def vlayer(p, gated):
        ga = torch.tanh(sd[p + "gate_attn"].float()) if gated else 1.0
        gf = torch.tanh(sd[p + "gate_ffn"].float()) if gated else 1.0
        qkv = torch.cat([sd[p + f"self_attn.{n}_proj.weight"] for n in ("q", "k", "v")], dim=0)
        return MllamaVisionLayer(
            ln1_w=d(sd[p + "input_layernorm.weight"]), ln1_b=d(sd[p + "input_layernorm.bias"]),
            ln2_w=d(sd[p + "post_attention_layernorm.weight"]), ln2_b=d(sd[p + "post_attention_layernorm.bias"]),
            qkv_w=d(qkv), o_w=d(sd[p + "self_attn.o_proj.weight"].float() * ga),
            fc1_w=d(sd[p + "mlp.fc1.weight"]), fc1_b=d(sd[p + "mlp.fc1.bias"]),
            fc2_w=d(sd[p + "mlp.fc2.weight"].float() * gf), fc2_b=d(sd[p + "mlp.fc2.bias"].float() * gf))

    w = MllamaDeviceWeights(
        patch_w=d(pad_cols(sd[V + "patch_embedding.weight"].reshape(E, -1), 64)),
        cls_pos=d(cls_pos), post_tile=d(post),
        ln_pre_w=d(sd[V + "layernorm_pre.weight"]), ln_pre_b=d(sd[V + "layernorm_pre.bias"]),
        ln_post_w=d(sd[V + "layernorm_post.weight"]), ln_post_b=d(sd[V + "layernorm_post.bias"]))
    w.v_layers = [vlayer(f"{V}transformer.layers.{i}.", False) for i in range(cfg.v_layers)]
    w.v_global = [vlayer(f"{V}global_transformer.layers.{i}.", True) for i in range(cfg.v_global_layers)]
    # projector: HF feature order is [final E | (e, i) interleaved]; the engine concatenates [final | inter_0 | inter_1 ...]
    ni = len(cfg.v_inter)
    pw = sd["model.multi_modal_projector.weight"].float()
    inter = pw[:, E:].view(cfg.hidden, E, ni).permute(0, 2, 1).reshape(cfg.hidden, E * ni)
    w.proj_w = d(torch.cat([pw[:, :E], inter], dim=1))
    w.proj_b = d(sd["model.multi_modal_projector.bias"])
    w.embed = d(sd[L + "embed_tokens.weight"])
    for i in range(cfg.layers):
        p = f"{L}layers.{i}."
        gu = interleave_gate_up(sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"])
        if i in cfg.cross_layers:
            ga = torch.tanh(sd[p + "cross_attn_attn_gate"].float())
            gm = torch.tanh(sd[p + "cross_attn_mlp_gate"].float())
            kv = torch.cat([sd[p + "cross_attn.k_proj.weight"], sd[p + "cross_attn.v_proj.weight"]], dim=0)
            w.layers.append(MllamaTextLayer(
                cross=True, ln1_w=d(sd[p + "input_layernorm.weight"]), ln2_w=d(sd[p + "post_attention_layernorm.weight"]),
                qkv_w=d(sd[p + "cross_attn.q_proj.weight"]), kv_w=d(kv),
                q_norm=d(sd[p + "cross_attn.q_norm.weight"]), k_norm=d(sd[p + "cross_attn.k_norm.weight"]),
                o_w=d(sd[p + "cross_attn.o_proj.weight"].float() * ga), gateup_w=d(gu),
                down_w=d(sd[p + "mlp.down_proj.weight"].float() * gm)))
        else:
            qkv = torch.cat([sd[p + f"self_attn.{n}_proj.weight"] for n in ("q", "k", "v")], dim=0)
            w.layers.append(MllamaTextLayer(
                cross=False, ln1_w=d(sd[p + "input_layernorm.weight"]), ln2_w=d(sd[p + "post_attention_layernorm.weight"]),
                qkv_w=d(qkv), kv_w=None, q_norm=None, k_norm=None, o_w=d(sd[p + "self_attn.o_proj.weight"]),
                gateup_w=d(gu), down_w=d(sd[p + "mlp.down_proj.weight"])))
    w.norm_w = d(sd[L + "norm.weight"])
    w.lm_head = d(sd["lm_head.weight"])
    return w


# ----------------------------------------------------------------------------- other weight sources
def random_device_weights(cfg: MllamaConfig, device, seed: int = 0, std: float = 0.02) -> MllamaDeviceWeights:
    """Seeded normal(0, std) bf16 weights generated ON the device at the exact shapes of ``cfg`` (throughput runs:
    no checkpoint exists offline; timing is valid, generated text is noise).  Norm weights are 1, gates tanh(0.5)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    bf = torch.bfloat16

    def rn(*shape, s=std):
        return (torch.randn(shape, generator=g, device=device, dtype=torch.float32) * s).to(bf)

    def ones(n):
        return torch.ones(n, dtype=bf, device=device)

    def zeros(n):
        return torch.zeros(n, dtype=bf, device=device)

    E, H, D = cfg.v_hidden, cfg.hidden, cfg.head_dim
    T, P, A = cfg.max_tiles, cfg.tile_tokens, cfg.max_ar_id + 1

    def vlayer():
        return MllamaVisionLayer(ones(E), zeros(E), ones(E), zeros(E), rn(3 * E, E), rn(E, E), rn(cfg.v_mlp, E),
                                 rn(cfg.v_mlp), rn(E, cfg.v_mlp), rn(E))

    w = MllamaDeviceWeights(patch_w=pad_cols(rn(E, 3 * cfg.patch * cfg.patch), 64), cls_pos=rn(A, T, P, E),
                            post_tile=rn(A, T, E), ln_pre_w=ones(E), ln_pre_b=zeros(E), ln_post_w=ones(E),
                            ln_post_b=zeros(E))
    w.v_layers = [vlayer() for _ in range(cfg.v_layers)]
    w.v_global = [vlayer() for _ in range(cfg.v_global_layers)]
    w.proj_w, w.proj_b = rn(H, cfg.v_out), rn(H)
    w.embed = rn(cfg.vocab + 8, H)
    for i in range(cfg.layers):
        if i in cfg.cross_layers:
            w.layers.append(MllamaTextLayer(True, ones(H), ones(H), rn(cfg.heads * D, H), rn(2 * cfg.kv_heads * D, H),
                                            ones(D), ones(D), rn(H, cfg.heads * D), rn(2 * cfg.intermediate, H),
                                            rn(H, cfg.intermediate)))
        else:
            w.layers.append(MllamaTextLayer(False, ones(H), ones(H), rn((cfg.heads + 2 * cfg.kv_heads) * D, H), None,
                                            None, None, rn(H, cfg.heads * D), rn(2 * cfg.intermediate, H),
                                            rn(H, cfg.intermediate)))
    w.norm_w = ones(H)
    w.lm_head = rn(cfg.vocab, H)
    return w


def config_from_hf_dir(path: str) -> MllamaConfig:
    """MllamaConfig from a LOCAL HF model directory's config.json (model_type "mllama")."""
    import json
    import os
    with open(os.path.join(path, "config.json")) as f:
        c = json.load(f)
    if c.get("model_type") != "mllama":
        raise ValueError(f"{path}: not an mllama checkpoint (model_type={c.get('model_type')!r})")
    t, v = c["text_config"], c["vision_config"]
    rope = t.get("rope_scaling") or t.get("rope_parameters") or {}
    eos = t.get("eos_token_id")
    if eos is None:
        eos = c.get("eos_token_id")
    gen_path = os.path.join(path, "generation_config.json")
    if eos is None and os.path.exists(gen_path):
        with open(gen_path) as f:
            eos = json.load(f).get("eos_token_id")
    if eos is None:
        eos = [128001, 128008, 128009]
    # preprocessor_config.json: the tile geometry must be the one the vision tower was built for
    pre, extra = {}, {}
    pre_path = os.path.join(path, "preprocessor_config.json")
    if os.path.exists(pre_path):
        with open(pre_path) as f:
            pre = json.load(f)
    size = pre.get("size") or {}
    if size.get("height") is not None and (int(size["height"]) != int(v["image_size"]) or int(size.get("width", size["height"])) != int(v["image_size"])):
        raise ValueError(f"{pre_path}: tile size {size} contradicts config.json image_size {v['image_size']}")
    if pre.get("max_image_tiles") is not None and int(pre["max_image_tiles"]) != int(v["max_num_tiles"]):
        raise ValueError(f"{pre_path}: max_image_tiles {pre['max_image_tiles']} contradicts config.json ({v['max_num_tiles']})")
    if pre.get("image_mean") is not None:
        extra["image_mean"] = tuple(float(x) for x in pre["image_mean"])
    if pre.get("image_std") is not None:
        extra["image_std"] = tuple(float(x) for x in pre["image_std"])
    return MllamaConfig(
        hidden=t["hidden_size"], layers=t["num_hidden_layers"], heads=t["num_attention_heads"],
        kv_heads=t["num_key_value_heads"], intermediate=t["intermediate_size"], vocab=t["vocab_size"],
        rms_eps=t.get("rms_norm_eps", 1e-5), rope_theta=t.get("rope_theta", rope.get("rope_theta", 500000.0)),
        rope_factor=rope.get("factor", 8.0), rope_low_freq=rope.get("low_freq_factor", 1.0),
        rope_high_freq=rope.get("high_freq_factor", 4.0),
        rope_orig_ctx=rope.get("original_max_position_embeddings", 8192),
        cross_layers=tuple(t["cross_attention_layers"]), image_token_id=c.get("image_token_index", 128256),
        eos_ids=tuple(eos) if isinstance(eos, (list, tuple)) else (eos,),
        v_hidden=v["hidden_size"], v_heads=v["attention_heads"], v_layers=v["num_hidden_layers"],
        v_global_layers=v["num_global_layers"], v_mlp=v["intermediate_size"],
        v_inter=tuple(v["intermediate_layers_indices"]), v_eps=v.get("norm_eps", 1e-5), image_size=v["image_size"],
        patch=v["patch_size"], max_tiles=v["max_num_tiles"], name=os.path.basename(os.path.normpath(path)), **extra)


def load_safetensors_dir(cfg: MllamaConfig, path: str, device) -> MllamaDeviceWeights:
    """Load ``*.safetensors`` shards from a LOCAL model directory (no hub access, ever)."""
    import glob
    import os
    from safetensors import safe_open
    files = sorted(glob.glob(os.path.join(path, "*.safetensors")))
    if not files:
        raise FileNotFoundError(f"no *.safetensors under {path} (the local backend only loads local files)")
    sd: Dict[str, torch.Tensor] = {}
    for fpath in files:
        with safe_open(fpath, framework="pt", device="cpu") as f:
            for k in f.keys():
                name = k
                if not k.startswith("model.") and not k.startswith("lm_head"):   # older checkpoints: no "model." prefix
                    name = "model." + k
                if name.startswith("model.language_model.model."):
                    name = "model.language_model." + name[len("model.language_model.model."):]
                if name == "model.language_model.lm_head.weight":
                    name = "lm_head.weight"
                sd[name] = f.get_tensor(k)
    return pack_device_weights(cfg, sd, device)
