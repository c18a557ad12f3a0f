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
