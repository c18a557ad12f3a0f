"""Configuration surface of the local MI355X backend.

Two things live here:

1. ``Qwen2VLConfig`` - the architecture constants of the Inspector model
   (Qwen2-VL-7B-Instruct ``config.json`` values) plus a tiny
   kernel-compatible variant used by the parity tests.
2. The reference's *model surface* (boundary B4): the ``Config`` field names of
   ``utils/config.py:42-76,:184`` (``vlm_inspector_model`` ...) read from the same
   environment variables, and the ``config/models.yaml:5-18`` schema
   (``inspector/auditor: {model_id, temperature, max_tokens, description, provider}``).
   A new ``provider`` value, ``"mi355x"``, selects this backend and ``model_id`` may
   then be a local model directory (or ``synthetic:<name>`` for seeded random weights).
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, field, asdict
from typing import Any, Dict, Optional, Tuple

LOCAL_PROVIDER = "mi355x"


@dataclass(frozen=True)
class Qwen2VLConfig:
    name: str = "qwen2-vl-7b"
    # text decoder
    hidden: int = 3584
    layers: int = 28
    heads: int = 28
    kv_heads: int = 4
    intermediate: int = 18944
    vocab: int = 152064
    rms_eps: float = 1e-6
    rope_theta: float = 1e6
    mrope_section: Tuple[int, int, int] = (16, 24, 24)
    # vision tower
    v_depth: int = 32
    v_embed: int = 1280
    v_heads: int = 16
    v_mlp: int = 5120
    patch: int = 14
    temporal: int = 2
    merge: int = 2
    min_pixels: int = 56 * 56
    max_pixels: int = 28 * 28 * 1280
    # normalisation constants of the image processor (CLIP's; a checkpoint's preprocessor_config.json may override them)
    image_mean: Tuple[float, float, float] = (0.48145466, 0.4578275, 0.40821073)
    image_std: Tuple[float, float, float] = (0.26862954, 0.26130258, 0.27577711)
    # vision tower family: "qwen2_vl" (LayerNorm, fc1/QuickGELU/fc2, full attention per image) or "qwen2_5_vl" (the
    # reference's code default, utils/config.py:42-45: RMSNorm, SwiGLU MLP with biases of width v_mlp, attention inside
    # v_window-pixel windows except in the v_fullatt blocks; TF:models/qwen2_5_vl/modeling_qwen2_5_vl.py:294-472)
    vision_arch: str = "qwen2_vl"
    v_window: int = 112
    v_fullatt: Tuple[int, ...] = (7, 15, 23, 31)
    # special tokens
    image_token_id: int = 151655
    vision_start_id: int = 151652
    vision_end_id: int = 151653
    eos_ids: Tuple[int, ...] = (151645, 151643)

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads

    @property
    def v_head_dim(self) -> int:
        return self.v_embed // self.v_heads

    @property
    def patch_dim(self) -> int:
        return 3 * self.temporal * self.patch * self.patch

    @property
    def v_mlp_pad(self) -> int:
        """MLP width as the kernels see it: Qwen2.5-VL's 3420 is zero-padded to a multiple of 64 (GEMM K-step of the down
        projection; also a multiple of 16 for the gate/up interleave).  Exact: padded gate/up rows are zero, so are
        their SwiGLU outputs and the padded down-projection columns."""
        return (self.v_mlp + 63) // 64 * 64 if self.vision_arch == "qwen2_5_vl" else self.v_mlp

    def validate_for_kernels(self) -> None:
        """The gfx950 kernels are specialised; refuse shapes they do not cover."""
        problems = []
        if self.head_dim != 128:
            problems.append(f"LLM head_dim must be 128, got {self.head_dim}")
        if self.v_head_dim != 80:
            problems.append(f"ViT head_dim must be 80, got {self.v_head_dim}")
        if self.heads % self.kv_heads or self.heads // self.kv_heads not in (1, 2, 4, 7, 8):
            problems.append("heads/kv_heads must be one of 1, 2, 4, 7, 8 (instantiated GQA group sizes)")
        if self.vision_arch not in ("qwen2_vl", "qwen2_5_vl"):
            problems.append(f"unknown vision_arch {self.vision_arch!r}")
        if self.vision_arch == "qwen2_5_vl" and (self.v_window % (self.patch * self.merge) or self.v_window <= 0):
            problems.append("v_window must be a positive multiple of patch * merge")
        for nm, v in (("hidden", self.hidden), ("intermediate", self.intermediate), ("v_embed", self.v_embed),
                      ("v_mlp", self.v_mlp_pad), ("merger", self.v_embed * self.merge ** 2)):
            if v % 64:
                problems.append(f"{nm}={v} must be a multiple of 64 (GEMM K-step)")
        if self.intermediate % 16 or self.vocab % 4:
            problems.append("intermediate % 16 and vocab % 4 required")
        if sum(self.mrope_section) * 2 != self.head_dim:
            problems.append("mrope_section must sum to head_dim/2")
        if self.hidden > 5120 or self.v_embed > 5120:
            problems.append("norm kernels support rows up to 5120")
        if problems:
            raise ValueError("unsupported model shape for the gfx950 kernels: " + "; ".join(problems))

    @classmethod
    def qwen2_vl_7b(cls) -> "Qwen2VLConfig":
        return cls()

    @classmethod
    def qwen2_5_vl_7b(cls) -> "Qwen2VLConfig":
        """Qwen2.5-VL-7B-Instruct: the text decoder has the Qwen2-VL-7B shapes; the tower is the windowed one."""
        return cls(name="qwen2.5-vl-7b", vision_arch="qwen2_5_vl", v_mlp=3420, v_window=112, v_fullatt=(7, 15, 23, 31))

    @classmethod
    def tiny_2_5(cls) -> "Qwen2VLConfig":
        """Kernel-compatible miniature of the Qwen2.5-VL tower: 3 blocks (block 1 full attention), 56-pixel windows
        (2 x 2 merged tokens = 16 patches), SwiGLU width 428 (padded to 448 like 3420 -> 3456)."""
        return cls(name="qwen2.5-vl-tiny", hidden=256, layers=2, heads=2, kv_heads=1, intermediate=704, vocab=512,
                   v_depth=3, v_embed=320, v_heads=4, v_mlp=428, image_token_id=500, vision_start_id=501,
                   vision_end_id=502, eos_ids=(503,), vision_arch="qwen2_5_vl", v_window=56, v_fullatt=(1,))

    @classmethod
    def tiny(cls) -> "Qwen2VLConfig":
        """Smallest shape every kernel specialisation accepts (parity tests, smoke)."""
        return cls(name="qwen2-vl-tiny", hidden=256, layers=2, heads=2, kv_heads=1, intermediate=704, vocab=512,
                   v_depth=2, v_embed=320, v_heads=4, v_mlp=1280, image_token_id=500, vision_start_id=501,
                   vision_end_id=502, eos_ids=(503,))

    @classmethod
    def from_hf_dir(cls, path: str) -> "Qwen2VLConfig":
        """Read a local HuggingFace ``config.json`` (Qwen2-VL or Qwen2.5-VL layout, flat or nested text_config)."""
        with open(os.path.join(path, "config.json")) as f:
            c = json.load(f)
        t = c.get("text_config", c)
        v = c.get("vision_config", {})
        arch = "qwen2_5_vl" if c.get("model_type") == "qwen2_5_vl" else "qwen2_vl"
        if arch == "qwen2_5_vl":        # TF:models/qwen2_5_vl/configuration_qwen2_5_vl.py: own key names
            v = dict(v)
            v.setdefault("embed_dim", v.get("hidden_size", 1280))
            vis_extra = dict(vision_arch=arch, v_mlp=int(v.get("intermediate_size", 3420)),
                             v_window=int(v.get("window_size", 112)),
                             v_fullatt=tuple(v.get("fullatt_block_indexes", (7, 15, 23, 31))))
        else:
            vis_extra = dict(v_mlp=int(v.get("embed_dim", 1280) * v.get("mlp_ratio", 4)))
        rope = t.get("rope_scaling") or t.get("rope_parameters") or c.get("rope_scaling") or {}
        # end-of-sequence ids: config.json (top level or text_config; either may be null), then generation_config.json
        eos = c.get("eos_token_id")
        if eos is None:
            eos = t.get("eos_token_id")
        gen_path = os.path.join(path, "generation_config.json")
        if os.path.exists(gen_path):
            with open(gen_path) as f:
                g_eos = json.load(f).get("eos_token_id")
            if g_eos is not None:
                as_list = lambda e: list(e) if isinstance(e, (list, tuple)) else [int(e)]      # noqa: E731
                eos = as_list(g_eos) + [e for e in (as_list(eos) if eos is not None else []) if e not in as_list(g_eos)]
        if eos is None:
            eos = 151645
        eos_ids = tuple(int(e) for e in eos) if isinstance(eos, (list, tuple)) else (int(eos),)
        if 151643 not in eos_ids and t.get("vocab_size", 0) > 151643:
            eos_ids = eos_ids + (151643,)
        # preprocessor_config.json (the checkpoint's image processor): pixel bounds of smart_resize and the normalisation
        # constants.  Both spellings occur: min_pixels / max_pixels, and size = {shortest_edge, longest_edge}
        # (TF:models/qwen2_vl/image_processing_qwen2_vl.py).  A released checkpoint's bounds decide how many image tokens
        # a frame becomes - reading them keeps that equal to what the published model runs.
        pre = {}
        pre_path = os.path.join(path, "preprocessor_config.json")
        if os.path.exists(pre_path):
            with open(pre_path) as f:
                pre = json.load(f)
        size = pre.get("size") or {}
        pix = {}
        mn = pre.get("min_pixels", size.get("shortest_edge"))
        mx = pre.get("max_pixels", size.get("longest_edge"))
        if mn is not None:
            pix["min_pixels"] = int(mn)
        if mx is not None:
            pix["max_pixels"] = int(mx)
        if pre.get("image_mean") is not None:
            pix["image_mean"] = tuple(float(x) for x in pre["image_mean"])
        if pre.get("image_std") is not None:
            pix["image_std"] = tuple(float(x) for x in pre["image_std"])
        for key, mine in (("patch_size", v.get("patch_size", 14)), ("merge_size", v.get("spatial_merge_size", 2)),
                          ("temporal_patch_size", v.get("temporal_patch_size", 2))):
            if pre.get(key) is not None and int(pre[key]) != int(mine):
                raise ValueError(f"{pre_path}: {key} = {pre[key]} contradicts config.json ({mine})")
        return cls(
            name=os.path.basename(os.path.normpath(path)),
            hidden=t["hidden_size"], layers=t["num_hidden_layers"], heads=t["num_attention_heads"],
            kv_heads=t["num_key_value_heads"], intermediate=t["intermediate_size"], vocab=t["vocab_size"],
            rms_eps=t.get("rms_norm_eps", 1e-6), rope_theta=t.get("rope_theta", rope.get("rope_theta", 1e6)),
            mrope_section=tuple(rope.get("mrope_section", (16, 24, 24))),
            v_depth=v.get("depth", 32), v_embed=v.get("embed_dim", 1280), v_heads=v.get("num_heads", 16),
            patch=v.get("patch_size", 14), **vis_extra,
            temporal=v.get("temporal_patch_size", 2), merge=v.get("spatial_merge_size", 2),
            image_token_id=c.get("image_token_id", 151655), vision_start_id=c.get("vision_start_token_id", 151652),
            vision_end_id=c.get("vision_end_token_id", 151653), eos_ids=eos_ids, **pix)


# ----------------------------------------------------------------------------- B4: reference Config surface
@dataclass
class AgentModelSettings:
    """One block of config/models.yaml (config/models.yaml:5-18)."""
    model_id: str
    temperature: float
    max_tokens: int
    description: str = ""
    provider: str = "huggingface"


def _env(name: str, default, cast):
    v = os.environ.get(name)
    if v is None or v == "":
        return default
    return cast(v)


@dataclass
class Config:
    """Subset of the reference's pydantic ``Config`` that the hot path reads (utils/config.py:42-76,:184).

    Field names and environment aliases are the reference's; defaults are the reference's defaults.
    """
    vlm_inspector_model: str = "Qwen/Qwen2.5-VL-7B-Instruct"
    vlm_inspector_temperature: float = 0.1
    vlm_inspector_max_tokens: int = 2048
    vlm_inspector_provider: str = "huggingface"
    vlm_auditor_model: str = "Qwen/Qwen2.5-VL-7B-Instruct"
    vlm_auditor_temperature: float = 0.2
    vlm_auditor_max_tokens: int = 2048
    vlm_auditor_provider: str = "huggingface"
    max_image_dimension: int = 2048
    log_level: str = "INFO"
    max_defects_auto: int = 2
    high_criticality_requires_review: bool = True
    huggingface_api_key: Optional[str] = None
    groq_api_key: Optional[str] = None

    @classmethod
    def from_env(cls) -> "Config":
        c = cls()
        c.vlm_inspector_model = _env("VLM_INSPECTOR_MODEL", c.vlm_inspector_model, str)
        c.vlm_inspector_temperature = _env("VLM_INSPECTOR_TEMPERATURE", c.vlm_inspector_temperature, float)
        c.vlm_inspector_max_tokens = _env("VLM_INSPECTOR_MAX_TOKENS", c.vlm_inspector_max_tokens, int)
        c.vlm_inspector_provider = _env("VLM_INSPECTOR_PROVIDER", c.vlm_inspector_provider, str)
        c.vlm_auditor_model = _env("VLM_AUDITOR_MODEL", c.vlm_auditor_model, str)
        c.vlm_auditor_temperature = _env("VLM_AUDITOR_TEMPERATURE", c.vlm_auditor_temperature, float)
        c.vlm_auditor_max_tokens = _env("VLM_AUDITOR_MAX_TOKENS", c.vlm_auditor_max_tokens, int)
        c.vlm_auditor_provider = _env("VLM_AUDITOR_PROVIDER", c.vlm_auditor_provider, str)
        c.max_image_dimension = _env("MAX_IMAGE_DIMENSION", c.max_image_dimension, int)
        c.log_level = _env("LOG_LEVEL", c.log_level, str)
        c.max_defects_auto = _env("MAX_DEFECTS_AUTO", c.max_defects_auto, int)
        c.huggingface_api_key = _env("HUGGINGFACE_API_KEY", None, str)
        c.groq_api_key = _env("GROQ_API_KEY", None, str)
        return c

    def apply_models_yaml(self, path: str) -> "Config":
        """Overlay a ``config/models.yaml`` file (the reference ships one but never loads it)."""
        for role, block in load_models_yaml(path).items():
            if role not in ("inspector", "auditor"):
                continue
            setattr(self, f"vlm_{role}_model", block.model_id)
            setattr(self, f"vlm_{role}_temperature", block.temperature)
            setattr(self, f"vlm_{role}_max_tokens", block.max_tokens)
            setattr(self, f"vlm_{role}_provider", block.provider)
        return self


def load_models_yaml(path: str) -> Dict[str, AgentModelSettings]:
    """Parse the reference's models.yaml schema; unknown top-level keys (``groq:`` ...) are ignored."""
    import yaml
    with open(path) as f:
        raw = yaml.safe_load(f) or {}
    out: Dict[str, AgentModelSettings] = {}
    for role in ("inspector", "auditor", "explainer"):
        b = raw.get(role)
        if not isinstance(b, dict):
            continue
        out[role] = AgentModelSettings(model_id=str(b["model_id"]), temperature=float(b.get("temperature", 0.1)),
                                       max_tokens=int(b.get("max_tokens", 1024)),
                                       description=str(b.get("description", "")),
                                       provider=str(b.get("provider", "huggingface")))
    return out


_config: Optional[Config] = None


def get_config() -> Config:
    """Process-global settings object, like the reference's ``utils.config.config`` singleton (:350).

    If the host application (the reference) is importable its own ``config`` object wins, so a
    drop-in deployment keeps one source of truth.
    """
    global _config
    if _config is None:
        try:  # pragma: no cover - only inside the reference application
            from utils.config import config as host_config  # type: ignore
            _config = host_config
        except Exception:
            _config = Config.from_env()
            yaml_path = os.environ.get("VIS_MODELS_YAML")
            if yaml_path:
                _config.apply_models_yaml(yaml_path)
    return _config


def set_config(cfg) -> None:
    global _config
    _config = cfg
