"""DEV-ONLY generator of the arithmetic golden vectors (G5/G6 of SURVEY.md section 8c).

Runs in the development container only (needs ``transformers`` 5.15.0, which is present there;
nothing here is imported by tests at run time).  It instantiates the *real*
``Qwen2VLForConditionalGeneration`` from a tiny config object (no hub access, no checkpoint),
loads the deterministic synthetic weights of ``vision_inspection_system_amd.weights.synth_state_dict``
and records, in fp32 on CPU:

  qwen2vl_tiny.npz   case A: one 56x84 image;  case B: two images (56x56 + 84x56) in one prompt (varlen ViT)
                     per case: input ids, pixel_values checksum + sample, merged image embeddings,
                     position ids, first-step logits, 16 greedy tokens
  smart_resize.json  smart_resize(h, w) for ~45 sizes and the HF PIL processor's patch layout checksum

Usage:  python tests/golden/gen_qwen2vl_golden.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from vision_inspection_system_amd.config import Qwen2VLConfig  # noqa: E402
from vision_inspection_system_amd.weights import synth_state_dict  # noqa: E402


def build_hf_model(cfg: Qwen2VLConfig, sd):
    from transformers import Qwen2VLConfig as HFConfig, Qwen2VLForConditionalGeneration
    hf_cfg = HFConfig(
        text_config=dict(hidden_size=cfg.hidden, num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads,
                         num_key_value_heads=cfg.kv_heads, intermediate_size=cfg.intermediate, vocab_size=cfg.vocab,
                         rms_norm_eps=cfg.rms_eps, max_position_embeddings=4096, tie_word_embeddings=False,
                         rope_parameters=dict(rope_type="default", rope_theta=cfg.rope_theta,
                                              mrope_section=list(cfg.mrope_section)),
                         bos_token_id=None, eos_token_id=None, pad_token_id=None),
        vision_config=dict(depth=cfg.v_depth, embed_dim=cfg.v_embed, num_heads=cfg.v_heads, hidden_size=cfg.hidden,
                           mlp_ratio=cfg.v_mlp // cfg.v_embed, patch_size=cfg.patch,
                           temporal_patch_size=cfg.temporal, spatial_merge_size=cfg.merge),
        image_token_id=cfg.image_token_id, video_token_id=cfg.image_token_id + 10,
        vision_start_token_id=cfg.vision_start_id, vision_end_token_id=cfg.vision_end_id,
        tie_word_embeddings=False)
    hf_cfg._attn_implementation = "eager"
    model = Qwen2VLForConditionalGeneration(hf_cfg).eval().float()
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
    from PIL import Image
    proc = Qwen2VLImageProcessorPil()
    feats = proc(images=[Image.fromarray(f) for f in frames], return_tensors="pt")
    pv, grid = feats["pixel_values"].float(), feats["image_grid_thw"]
    input_ids = torch.tensor([ids], dtype=torch.long)
    mm = (input_ids == cfg.image_token_id).int()
    with torch.no_grad():
        img = model.get_image_features(pv, grid).pooler_output
        img = torch.cat(list(img), dim=0)
        pos, _ = model.model.get_rope_index(input_ids, mm, image_grid_thw=grid)
        out = model(input_ids=input_ids, pixel_values=pv, image_grid_thw=grid, mm_token_type_ids=mm)
        logits = out.logits[0, -1].float()
        gen = model.generate(input_ids=input_ids, pixel_values=pv, image_grid_thw=grid, mm_token_type_ids=mm,
                             max_new_tokens=n_new, do_sample=False)
    return dict(pixel_values=pv.numpy(), grid=grid.numpy(), image_embeds=img.numpy(), position_ids=pos[:, 0].numpy(),
                first_logits=logits.numpy(), tokens=gen[0, len(ids):].numpy())


def main():
    cfg = Qwen2VLConfig.tiny()
    sd = synth_state_dict(cfg, seed=0)
    model = build_hf_model(cfg, sd)
    rng = np.random.default_rng(2024)
    fa = rng.integers(0, 256, (56, 84, 3), dtype=np.uint8)
    fb1 = rng.integers(0, 256, (56, 56, 3), dtype=np.uint8)
    fb2 = rng.integers(0, 256, (84, 56, 3), dtype=np.uint8)

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
        out[f"{tag}_pixel_values_sum"] = np.array([c["pixel_values"].astype(np.float64).sum()])
        out[f"{tag}_pixel_values_rows"] = c["pixel_values"][[0, 5, -1]]
        out[f"{tag}_grid"] = c["grid"]
        out[f"{tag}_image_embeds"] = c["image_embeds"].astype(np.float32)
        out[f"{tag}_position_ids"] = c["position_ids"]
        out[f"{tag}_first_logits"] = c["first_logits"].astype(np.float32)
        out[f"{tag}_tokens"] = c["tokens"]
    np.savez_compressed(os.path.join(HERE, "qwen2vl_tiny.npz"), **out)

    from transformers.models.qwen2_vl.image_processing_pil_qwen2_vl import smart_resize
    sizes = [(448, 448), (1024, 1024), (2048, 2048), (56, 56), (28, 28), (30, 2000), (2000, 30), (1080, 1920),
             (1920, 1080), (333, 777), (14, 14), (100, 100), (57, 57), (4096, 4096), (980, 980), (1, 150),
             (640, 480), (480, 640), (3000, 4000), (27, 1000), (1000, 27), (55, 55), (84, 56), (56, 84)]
    rs = np.random.default_rng(7)
    sizes += [(int(a_), int(b_)) for a_, b_ in rs.integers(20, 3000, (21, 2))]
    table = [{"h": h, "w": w, "out": list(smart_resize(h, w))} for h, w in sizes]
    with open(os.path.join(HERE, "smart_resize.json"), "w") as f:
        json.dump(table, f)
    print("wrote qwen2vl_tiny.npz, smart_resize.json;", "tokens a:", a["tokens"].tolist(), "tokens b:", b["tokens"].tolist())


if __name__ == "__main__":
    main()
