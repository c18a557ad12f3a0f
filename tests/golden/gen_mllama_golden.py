"""DEV-ONLY generator of the mllama (row f2) golden vectors.

Runs in the development container only (needs ``transformers`` 5.15.0; nothing here is imported by tests at run
time).  It instantiates the *real* ``MllamaForConditionalGeneration`` and ``MllamaImageProcessorPil`` from a tiny
config (no hub access, no checkpoint), loads the deterministic synthetic weights of
``vision_inspection_system_amd.mllama_weights.synth_state_dict`` and records, in fp32 on CPU:

  mllama_tiny.npz   case A: 40x100 image (2 tiles), image token right after the header, then text
                    case B: 100x90 image (4 tiles), text first, image token last (the reference's part order)
                    case C: 30x30 image (1 tile, upscaled), image first
                    per case: image, input ids, processor output (pixel tiles checksum + sample, aspect ratio id,
                    tile count), cross-attention states, first-step logits, 12 greedy tokens
  plus canvas choices for ~40 image sizes (tile 560, 4 tiles) from the real processor helpers.

Usage:  python tests/golden/gen_mllama_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from vision_inspection_system_amd.mllama_weights import MllamaConfig, synth_state_dict  # noqa: E402

CLIP_MEAN = [0.48145466, 0.4578275, 0.40821073]
CLIP_STD = [0.26862954, 0.26130258, 0.27577711]


def build_hf_model(cfg: MllamaConfig, sd):
    from transformers import MllamaConfig as HFConfig, MllamaForConditionalGeneration
    ars = [[w, h] for w in range(1, cfg.max_tiles + 1) for h in range(1, cfg.max_tiles + 1) if w * h <= cfg.max_tiles]
    hf = HFConfig(
        vision_config=dict(hidden_size=cfg.v_hidden, attention_heads=cfg.v_heads, num_hidden_layers=cfg.v_layers,
                           num_global_layers=cfg.v_global_layers, intermediate_size=cfg.v_mlp,
                           intermediate_layers_indices=list(cfg.v_inter), image_size=cfg.image_size,
                           patch_size=cfg.patch, max_num_tiles=cfg.max_tiles, vision_output_dim=cfg.v_out,
                           supported_aspect_ratios=ars, norm_eps=cfg.v_eps),
        text_config=dict(hidden_size=cfg.hidden, num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads,
                         num_key_value_heads=cfg.kv_heads, intermediate_size=cfg.intermediate, vocab_size=cfg.vocab,
                         cross_attention_layers=list(cfg.cross_layers), rms_norm_eps=cfg.rms_eps,
                         max_position_embeddings=4096,
                         rope_parameters=dict(rope_type="llama3", rope_theta=cfg.rope_theta, factor=cfg.rope_factor,
                                              low_freq_factor=cfg.rope_low_freq, high_freq_factor=cfg.rope_high_freq,
                                              original_max_position_embeddings=cfg.rope_orig_ctx),
                         pad_token_id=cfg.vocab - 1, bos_token_id=1, eos_token_id=2, tie_word_embeddings=False),
        image_token_index=cfg.image_token_id)
    hf._attn_implementation = "eager"
    model = MllamaForConditionalGeneration(hf).eval().float()
    own = model.state_dict()
    missing = [k for k in own if k not in sd]
    extra = [k for k in sd if k not in own]
    if missing or extra:
        raise KeyError(f"state dict mismatch: missing {missing[:4]} extra {extra[:4]}")
    model.load_state_dict({k: v.reshape(own[k].shape) for k, v in sd.items()}, strict=True)
    return model


def main():
    from transformers.models.mllama.image_processing_pil_mllama import (MllamaImageProcessorPil,
                                                                          get_image_size_fit_to_canvas,
                                                                          get_optimal_tiled_canvas)
    cfg = MllamaConfig.tiny()
    sd = synth_state_dict(cfg, seed=0)
    model = build_hf_model(cfg, sd)
    proc = MllamaImageProcessorPil(size={"height": cfg.image_size, "width": cfg.image_size},
                                   max_image_tiles=cfg.max_tiles, image_mean=CLIP_MEAN, image_std=CLIP_STD)
    rng = np.random.default_rng(11)
    out = {}
    cases = {
        "a": ((40, 100), "image_first"),
        "b": ((100, 90), "image_last"),
        "c": ((30, 30), "image_first"),
    }
    for name, ((h, w), order) in cases.items():
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        text = rng.integers(3, cfg.vocab - 8, 9).tolist()
        ids = [1, 5, 6] + ([cfg.image_token_id] + text if order == "image_first" else text + [cfg.image_token_id]) + [7, 8]
        enc = proc.preprocess([[img]], return_tensors="pt")
        pv = enc["pixel_values"].float()                       # [1,1,T,3,S,S]
        n_tiles = int(enc["num_tiles"][0][0])
        ar_id = int(enc["aspect_ratio_ids"][0, 0])
        S = len(ids)
        loc = ids.index(cfg.image_token_id)
        n_new = 12
        xmask = torch.zeros(1, S + n_new, 1, cfg.max_tiles)
        xmask[0, loc:, 0, :n_tiles] = 1
        with torch.no_grad():
            vis = model.model.vision_model(pixel_values=pv, aspect_ratio_ids=enc["aspect_ratio_ids"],
                                           aspect_ratio_mask=enc["aspect_ratio_mask"]).last_hidden_state
            cross = model.model.multi_modal_projector(vis).reshape(-1, cfg.hidden)
            o = model(input_ids=torch.tensor([ids]), pixel_values=pv, aspect_ratio_ids=enc["aspect_ratio_ids"],
                      aspect_ratio_mask=enc["aspect_ratio_mask"], cross_attention_mask=xmask[:, :S], use_cache=True)
            logits0 = o.logits[0, -1].float()
            past = o.past_key_values
            toks, logit_steps = [], [logits0]
            cur = int(torch.argmax(logits0))
            toks.append(cur)
            for step in range(1, n_new):
                o = model(input_ids=torch.tensor([[cur]]), past_key_values=past,
                          cross_attention_mask=xmask[:, :S + step], use_cache=True)
                lg = o.logits[0, -1].float()
                logit_steps.append(lg)
                past = o.past_key_values
                cur = int(torch.argmax(lg))
                toks.append(cur)
        out[f"{name}_image"] = img
        out[f"{name}_ids"] = np.array(ids, dtype=np.int64)
        out[f"{name}_n_tiles"] = np.array(n_tiles)
        out[f"{name}_ar_id"] = np.array(ar_id)
        out[f"{name}_pixel_sum"] = pv.double().sum().numpy()
        out[f"{name}_pixel_sample"] = pv.reshape(-1)[::97].numpy()
        out[f"{name}_cross_states"] = cross.numpy()
        out[f"{name}_logits"] = torch.stack(logit_steps).numpy()
        out[f"{name}_tokens"] = np.array(toks, dtype=np.int64)
        print(name, "tiles", n_tiles, "ar", ar_id, "S", S, "tokens", toks)
    sizes = []
    for (h, w) in [(1024, 1024), (480, 640), (640, 480), (100, 2000), (2000, 100), (560, 560), (561, 560), (300, 300),
                   (1120, 1120), (1121, 1120), (700, 1500), (1500, 700), (50, 50), (1, 1), (559, 1121), (2240, 560),
                   (333, 777), (777, 333), (1680, 560), (560, 1680), (900, 900), (2048, 2048), (768, 1024), (37, 53)]:
        ch, cw = get_optimal_tiled_canvas(h, w, 4, 560)
        nh, nw = get_image_size_fit_to_canvas(h, w, ch, cw, 560)
        sizes.append([h, w, int(ch), int(cw), int(nh), int(nw)])
    out["canvas_cases"] = np.array(sizes, dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "mllama_tiny.npz"), **out)
    print("wrote", os.path.join(HERE, "mllama_tiny.npz"))


if __name__ == "__main__":
    main()
