"""DEV-ONLY generator of the Qwen2.5-VL golden vectors (windowed vision tower; the reference's code-default model family,
/root/reference utils/config.py:42-45).  Instantiates the real ``Qwen2_5_VLForConditionalGeneration`` of transformers
5.15.0 from a tiny config object (no hub access, no checkpoint), loads the deterministic synthetic weights of
``weights.synth_state_dict(Qwen2VLConfig.tiny_2_5())`` and records, in fp32 on CPU, per case: the window permutation,
merged image embeddings, M-RoPE position ids, first-step logits and 16 greedy tokens.

  case A: one 112x140 image (8x10 patches -> 4x5 merged tokens: 2x2-token windows, ragged right edge)
  case B: two images (84x84 and 56x112) in one prompt

Usage:  python tests/golden/gen_qwen25vl_golden.py   (writes qwen25vl_tiny.npz)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from vision_inspection_system_amd.config import Qwen2VLConfig  # noqa: E402
from vision_inspection_system_amd.weights import synth_state_dict  # noqa: E402


def build_hf_model(cfg, sd):
    from transformers import Qwen2_5_VLConfig as HFConfig, Qwen2_5_VLForConditionalGeneration
    hf_cfg = HFConfig(
        text_config=dict(hidden_size=cfg.hidden, num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads,
                         num_key_value_heads=cfg.kv_heads, intermediate_size=cfg.intermediate, vocab_size=cfg.vocab,
                         rms_norm_eps=cfg.rms_eps, max_position_embeddings=4096, tie_word_embeddings=False,
                         rope_parameters=dict(rope_type="default", rope_theta=cfg.rope_theta,
                                              mrope_section=list(cfg.mrope_section)),
                         bos_token_id=None, eos_token_id=None, pad_token_id=None, use_sliding_window=False),
        vision_config=dict(depth=cfg.v_depth, hidden_size=cfg.v_embed, num_heads=cfg.v_heads,
                           intermediate_size=cfg.v_mlp, out_hidden_size=cfg.hidden, patch_size=cfg.patch,
                           temporal_patch_size=cfg.temporal, spatial_merge_size=cfg.merge, window_size=cfg.v_window,
                           fullatt_block_indexes=list(cfg.v_fullatt), hidden_act="silu"),
        image_token_id=cfg.image_token_id, video_token_id=cfg.image_token_id + 10,
        vision_start_token_id=cfg.vision_start_id, vision_end_token_id=cfg.vision_end_id,
        tie_word_embeddings=False)
    hf_cfg._attn_implementation = "eager"
    model = Qwen2_5_VLForConditionalGeneration(hf_cfg).eval().float()
    own = model.state_dict()
    mapped = {}
    for k, v in sd.items():
        cands = [k, "model." + k if k.startswith("visual.") else k,
                 k.replace("model.", "model.language_model.", 1) if k.startswith("model.") else k]
        hit = [c for c in cands if c in own]
        if not hit:
            raise KeyError(f"no HF parameter for {k}; HF keys look like {list(own)[:5]}")
        mapped[hit[0]] = v.reshape(own[hit[0]].shape)
    missing = [k for k in own if k not in mapped and "inv_freq" not in k]
    if missing:
        raise KeyError(f"HF parameters not covered: {missing[:5]}")
    model.load_state_dict(mapped, strict=False)
    return model


def run_case(model, cfg, frames, ids, n_new=16):
    from transformers.models.qwen2_vl.image_processing_pil_qwen2_vl import Qwen2VLImageProcessorPil
    from transformers.vision_utils import get_vision_window_index
    from PIL import Image
    proc = Qwen2VLImageProcessorPil()
    feats = proc(images=[Image.fromarray(f) for f in frames], return_tensors="pt")
    pv, grid = feats["pixel_values"].float(), feats["image_grid_thw"]
    input_ids = torch.tensor([ids], dtype=torch.long)
    mm = (input_ids == cfg.image_token_id).int()
    with torch.no_grad():
        widx, cu = get_vision_window_index(grid, spatial_merge_size=cfg.merge, window_size=cfg.v_window, patch_size=cfg.patch)
        img = model.get_image_features(pv, grid).pooler_output
        img = torch.cat(list(img), dim=0)
        pos, _ = model.model.get_rope_index(input_ids, mm, image_grid_thw=grid)
        out = model(input_ids=input_ids, pixel_values=pv, image_grid_thw=grid, mm_token_type_ids=mm)
        logits = out.logits[0, -1].float()
        gen = model.generate(input_ids=input_ids, pixel_values=pv, image_grid_thw=grid, mm_token_type_ids=mm,
                             max_new_tokens=n_new, do_sample=False)
    return dict(grid=grid.numpy(), window_index=widx.numpy(), cu_window=cu.numpy(), image_embeds=img.numpy(),
                position_ids=pos[:, 0].numpy(), first_logits=logits.numpy(), tokens=gen[0, len(ids):].numpy())


def main():
    cfg = Qwen2VLConfig.tiny_2_5()
    sd = synth_state_dict(cfg, seed=0)
    model = build_hf_model(cfg, sd)
    rng = np.random.default_rng(2025)
    fa = rng.integers(0, 256, (112, 140, 3), dtype=np.uint8)
    fb1 = rng.integers(0, 256, (84, 84, 3), dtype=np.uint8)
    fb2 = rng.integers(0, 256, (56, 112, 3), dtype=np.uint8)

    def prompt(frames, text_a, text_b):
        ids = [256] + text_a
        for f in frames:
            n = (f.shape[0] // 14) * (f.shape[1] // 14) // 4
            ids += [cfg.vision_start_id] + [cfg.image_token_id] * n + [cfg.vision_end_id]
        return ids + text_b

    ids_a = prompt([fa], [10, 11, 12], [20, 21, 22, 23, 24])
    ids_b = prompt([fb1, fb2], [65, 66], [97, 98, 99, 100])
    a = run_case(model, cfg, [fa], ids_a)
    b = run_case(model, cfg, [fb1, fb2], ids_b)
    out = {"frame_a": fa, "frame_b1": fb1, "frame_b2": fb2, "ids_a": np.array(ids_a), "ids_b": np.array(ids_b)}
    for tag, c in (("a", a), ("b", b)):
        for k in ("grid", "window_index", "cu_window", "position_ids", "tokens"):
            out[f"{tag}_{k}"] = c[k]
        out[f"{tag}_image_embeds"] = c["image_embeds"].astype(np.float32)
        out[f"{tag}_first_logits"] = c["first_logits"].astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "qwen25vl_tiny.npz"), **out)
    print("wrote qwen25vl_tiny.npz; tokens a:", a["tokens"].tolist(), "tokens b:", b["tokens"].tolist(),
          "windows a:", a["cu_window"].tolist())


if __name__ == "__main__":
    main()
