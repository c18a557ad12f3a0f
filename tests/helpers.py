"""Shared test helpers (oracle config mapping, tiny-model fixtures)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def ref_config(cfg):
    from oracle import qwen2vl_ref as R
    return R.RefConfig(hidden=cfg.hidden, layers=cfg.layers, heads=cfg.heads, kv_heads=cfg.kv_heads,
                       intermediate=cfg.intermediate, vocab=cfg.vocab, rms_eps=cfg.rms_eps,
                       rope_theta=cfg.rope_theta, mrope_section=tuple(cfg.mrope_section), v_depth=cfg.v_depth,
                       v_embed=cfg.v_embed, v_heads=cfg.v_heads, v_mlp=cfg.v_mlp, patch=cfg.patch,
                       temporal=cfg.temporal, merge=cfg.merge, image_token_id=cfg.image_token_id)


def oracle_inputs(frames):
    """uint8 frames -> (pixel_values tensor, grids) through the oracle's preprocessing."""
    from oracle import qwen2vl_ref as R
    pvs, grids = [], []
    for f in frames:
        pv, g = R.preprocess_u8(f)
        pvs.append(pv)
        grids.append(g)
    return torch.from_numpy(np.concatenate(pvs, axis=0)), grids


def load_golden():
    return np.load(os.path.join(GOLDEN, "qwen2vl_tiny.npz"))


# ----------------------------------------------------------------------------- a3 fixtures: seeded image recipes
ENCODE_RECIPES = [
    # name, PIL mode, (w, h), seed, file format
    {"name": "rgb_small", "mode": "RGB", "size": [120, 90], "seed": 1, "format": "PNG"},
    {"name": "rgb_448", "mode": "RGB", "size": [448, 448], "seed": 2, "format": "PNG"},
    {"name": "rgba", "mode": "RGBA", "size": [200, 150], "seed": 3, "format": "PNG"},
    {"name": "palette", "mode": "P", "size": [160, 160], "seed": 4, "format": "PNG"},
    {"name": "gray_alpha", "mode": "LA", "size": [96, 64], "seed": 5, "format": "PNG"},
    {"name": "gray", "mode": "L", "size": [77, 131], "seed": 6, "format": "PNG"},
    {"name": "wide_over_max", "mode": "RGB", "size": [2500, 700], "seed": 7, "format": "PNG"},
    {"name": "tall_over_auditor_max", "mode": "RGB", "size": [600, 1500], "seed": 8, "format": "PNG"},
    {"name": "jpeg_source", "mode": "RGB", "size": [320, 240], "seed": 9, "format": "JPEG"},
    {"name": "noise_2048", "mode": "RGB", "size": [2048, 2048], "seed": 10, "format": "PNG"},
    {"name": "smooth_1024", "mode": "RGB", "size": [1024, 1024], "seed": 11, "format": "PNG", "smooth": True},
]


LARGE_RECIPES = [   # encoded with max_size 6000 (no thumbnail): q85 > 5 MB -> the q60 retry; the second also > 10 MB at q60
    {"name": "noise_3400_q60", "mode": "RGB", "size": [3400, 3400], "seed": 12, "format": "PNG"},
    {"name": "noise_5200_refused", "mode": "RGB", "size": [5200, 5200], "seed": 13, "format": "PNG"},
]


def make_recipe_image(recipe: dict, path) -> None:
    """Write the image a recipe describes (seeded; identical in the generator and in the test)."""
    from PIL import Image
    w, h = recipe["size"]
    rng = np.random.default_rng(1000 + recipe["seed"])
    bands = {"RGB": 3, "RGBA": 4, "LA": 2, "L": 1, "P": 1}[recipe["mode"]]
    if recipe.get("smooth"):
        yy, xx = np.mgrid[0:h, 0:w]
        a = np.stack([(xx * 255 // max(w - 1, 1)), (yy * 255 // max(h - 1, 1)), ((xx + yy) % 256)], axis=-1).astype(np.uint8)
    else:
        a = rng.integers(0, 256, (h, w, bands), dtype=np.uint8)
    if recipe["mode"] == "P":
        img = Image.fromarray(a[:, :, 0], "P")
        img.putpalette(rng.integers(0, 256, 768, dtype=np.uint8).tobytes())
    elif bands == 1:
        img = Image.fromarray(a[:, :, 0], "L")
    else:
        img = Image.fromarray(a, recipe["mode"])
    if recipe["format"] == "JPEG":
        img.save(path, format="JPEG", quality=92)
    else:
        img.save(path, format="PNG")


# ----------------------------------------------------------------------------- token criterion (GPU tests)
def teacher_forced_parity(eng, first_logits, ref_toks, ref_logits, tol, use_graph=False):
    """EVERY generated step against the oracle, not only a prefix: after the prompt pass the engine is fed the ORACLE's
    token at each step (its own pick is overwritten in ``cur_token``), so a near-tie at step t cannot hide what
    happens at steps t+1.. .  Per step: logits within ``tol`` (absolute) of the oracle's, and the greedy pick equal to
    the oracle's unless the oracle's top-2 margin at that step is below 2 x tol (a genuine near-tie).
    Call right after ``eng.prefill``.  Returns the number of near-tie steps whose pick differed."""
    ties = 0
    logits = first_logits.float().cpu()
    for t in range(len(ref_toks)):
        ref = ref_logits[t].float()
        err = float((logits - ref).abs().max())
        assert err < tol, f"step {t}: logits differ from the oracle by {err:.4f} (tolerance {tol})"
        if int(logits.argmax()) != int(ref.argmax()):
            top2 = torch.topk(ref, 2).values
            margin = float(top2[0] - top2[1])
            assert margin < 2 * tol, f"step {t}: pick {int(logits.argmax())} vs oracle {int(ref.argmax())}, " \
                                     f"oracle margin {margin:.4f} is not a near-tie"
            ties += 1
        if t + 1 < len(ref_toks):
            eng.cur_token.fill_(int(ref_toks[t]))
            eng.decode(1, use_graph=use_graph)
            logits = eng.logits.float().cpu()
    return ties


def dequantised_sd(cfg, sd):
    """State dict whose LLM projections / lm_head are the engine's e4m3 weights, de-quantised (CPU, same quantiser)."""
    from vision_inspection_system_amd import hip
    from vision_inspection_system_amd.weights import interleave_gate_up

    def dq(w):
        q, s = hip.quantize_fp8_rows(w.to(torch.bfloat16))
        return q.view(torch.float8_e4m3fn).float() * s[:, None]

    dsd = dict(sd)
    for i in range(cfg.layers):
        p = f"model.layers.{i}."
        qkv = dq(torch.cat([sd[p + f"self_attn.{n}_proj.weight"] for n in ("q", "k", "v")], dim=0))
        nq, nk = cfg.heads * cfg.head_dim, cfg.kv_heads * cfg.head_dim
        dsd[p + "self_attn.q_proj.weight"], dsd[p + "self_attn.k_proj.weight"], dsd[p + "self_attn.v_proj.weight"] = \
            qkv[:nq], qkv[nq:nq + nk], qkv[nq + nk:]
        dsd[p + "self_attn.o_proj.weight"] = dq(sd[p + "self_attn.o_proj.weight"])
        gu = dq(interleave_gate_up(sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"]))
        gu = gu.view(cfg.intermediate // 16, 2, 16, cfg.hidden)
        dsd[p + "mlp.gate_proj.weight"] = gu[:, 0].reshape(cfg.intermediate, cfg.hidden)
        dsd[p + "mlp.up_proj.weight"] = gu[:, 1].reshape(cfg.intermediate, cfg.hidden)
        dsd[p + "mlp.down_proj.weight"] = dq(sd[p + "mlp.down_proj.weight"])
    dsd["lm_head.weight"] = dq(sd["lm_head.weight"])
    for i in range(cfg.v_depth):
        p = f"visual.blocks.{i}."
        for n in ("attn.qkv.weight", "attn.proj.weight", "mlp.fc1.weight", "mlp.fc2.weight"):
            dsd[p + n] = dq(sd[p + n])
    return dsd


# ----------------------------------------------------------------------------- real model directories
HF_DIRS = os.path.join(GOLDEN, "hf_dirs")


def materialize_hf_dir(family: str, dst) -> str:
    """Complete local HuggingFace model directory for ``family`` (qwen2vl_tiny / qwen25vl_tiny / mllama_tiny) under ``dst``:
    the committed files transformers 5.15 wrote (tests/golden/gen_hf_dir.py: config.json, generation_config.json,
    preprocessor_config.json, tokenizer.json, ...) plus model.safetensors REBUILT here from the seeded synth_state_dict under
    exactly the tensor names, shapes and dtype save_pretrained used (manifest.json) - the 26 MB file is not committed."""
    import json
    import shutil
    from safetensors.torch import save_file
    src = os.path.join(HF_DIRS, family)
    dst = str(dst)
    os.makedirs(dst, exist_ok=True)
    for f in os.listdir(src):
        if f not in ("manifest.json", "expected.npz"):
            shutil.copy(os.path.join(src, f), os.path.join(dst, f))
    with open(os.path.join(src, "manifest.json")) as f:
        manifest = json.load(f)
    if family == "mllama_tiny":
        from vision_inspection_system_amd.mllama_weights import MllamaConfig, synth_state_dict
        sd = synth_state_dict(MllamaConfig.tiny(), seed=0)

        def synth_name(k):       # the names save_pretrained writes (old checkpoint layout) -> the module names synth uses
            if k == "language_model.lm_head.weight":
                return "lm_head.weight"
            if k.startswith("language_model.model."):
                return "model.language_model." + k[len("language_model.model."):]
            return "model." + k
    else:
        from vision_inspection_system_amd.config import Qwen2VLConfig
        from vision_inspection_system_amd.weights import synth_state_dict
        cfg = Qwen2VLConfig.tiny() if family == "qwen2vl_tiny" else Qwen2VLConfig.tiny_2_5()
        sd = synth_state_dict(cfg, seed=0)

        def synth_name(k):
            return k
    out, used = {}, set()
    for k, meta in manifest.items():
        s = synth_name(k)
        if s not in sd:
            raise KeyError(f"{family}: save_pretrained wrote {k!r}; no synthetic tensor {s!r}")
        t = sd[s].reshape(meta["shape"]).to(getattr(torch, meta["dtype"])).contiguous()
        out[k] = t
        used.add(s)
    unused = [k for k in sd if k not in used]
    if unused:
        raise KeyError(f"{family}: synthetic tensors never written by save_pretrained: {unused[:4]}")
    save_file(out, os.path.join(dst, "model.safetensors"), metadata={"format": "pt"})
    return dst


def hf_expected(family: str):
    return np.load(os.path.join(HF_DIRS, family, "expected.npz"))
