"""Logit error against depth at the exact Qwen2-VL-7B shapes: the fp8 configuration (fp8 MFMA prompt
pass + e4m3 decode weights) against the bf16 engine on the SAME weights, for depth = 1, 2, 4, 8, 16, 28 decoder layers
(ViT blocks: min(depth, 32)), on three weight families:
  flat      normal(0, 0.02) everywhere - the throughput benchmark's weights
  scaled    variance-preserving (normal(0, 1 / fan_in)), as the one-layer oracle tests use
  scaled+bg scaled, with the matrices that write into the residual stream damped by 1 / sqrt(2 L) (GPT-2 style init of
            trained transformers)
Per depth: first-step logits and three teacher-forced decode steps - rms difference relative to the logits' standard
deviation and to their range, top-1 agreement, overlap of the top-5 sets.   python tools/depth_error.py [--depths 1,4,28]"""
import argparse
import dataclasses
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vision_inspection_system_amd.config import Qwen2VLConfig  # noqa: E402
from vision_inspection_system_amd.engine import Qwen2VLEngine  # noqa: E402
from vision_inspection_system_amd.weights import random_device_weights  # noqa: E402


def sliced(w, layers, v_depth):
    return dataclasses.replace(w, llm=w.llm[:layers], vit=w.vit[:v_depth])


def compare(a, b):
    a, b = a.float(), b.float()
    d = a - b
    sd, rng = float(b.std()), float(b.abs().max())
    rms = float(d.pow(2).mean().sqrt())
    t5a, t5b = set(torch.topk(a, 5).indices.tolist()), set(torch.topk(b, 5).indices.tolist())
    return {"rms/std": rms / sd, "rms/range": rms / rng, "max/range": float(d.abs().max()) / rng,
            "top1": int(a.argmax()) == int(b.argmax()), "top5": len(t5a & t5b)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--depths", default="1,2,4,8,16,28")
    ap.add_argument("--families", default="flat,scaled,scaled+bg")
    ap.add_argument("--steps", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    full = Qwen2VLConfig.qwen2_vl_7b()
    rng = np.random.default_rng(21)
    frame = torch.from_numpy(rng.integers(0, 256, (980, 980, 3), dtype=np.uint8)).to(dev)
    n_img = (980 // 14) ** 2 // 4
    text = rng.integers(0, 1000, 1024).tolist()
    ids = text[:960] + [full.vision_start_id] + [full.image_token_id] * n_img + [full.vision_end_id] + text[960:]
    for fam in a.families.split(","):
        kw = {"flat": {}, "scaled": {"scaled": True},
              "scaled+bg": {"scaled": True, "branch_gain": (2 * full.layers) ** -0.5}}[fam]
        w = random_device_weights(full, dev, seed=3, **kw)
        for depth in [int(x) for x in a.depths.split(",")]:
            cfg = dataclasses.replace(full, layers=min(depth, full.layers), v_depth=min(depth, full.v_depth))
            ws = sliced(w, cfg.layers, cfg.v_depth)
            e16 = Qwen2VLEngine(cfg, ws, dev, max_ctx=2560)
            e8 = Qwen2VLEngine(cfg, ws, dev, max_ctx=2560, prefill_dtype="fp8", decode_weights="fp8")
            rows = []
            t16, t8 = {}, {}
            e16.prefill(ids, [frame], taps=t16, max_new_tokens=8)
            e8.prefill(ids, [frame], taps=t8, max_new_tokens=8)
            rows.append(compare(t8["first_logits"], t16["first_logits"]))
            img = compare(t8["image_embeds"].flatten(), t16["image_embeds"].flatten())
            for _ in range(a.steps):
                tok = int(e16.logits.float().argmax())
                for e in (e16, e8):
                    e.cur_token.fill_(tok)
                    e.decode(1, use_graph=False)
                rows.append(compare(e8.logits, e16.logits))
            print(f"[depth] {fam:10s} L={cfg.layers:2d} V={cfg.v_depth:2d}  image feats rms/std {img['rms/std']:.4f} | "
                  + " | ".join(f"rms/std {r['rms/std']:.4f} rms/rng {r['rms/range']:.4f} max/rng {r['max/range']:.4f} "
                               f"top1 {int(r['top1'])} top5 {r['top5']}" for r in rows), flush=True)
            del e16, e8
            torch.cuda.empty_cache()
        del w
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
