"""Soak / race check at exact 7B shapes: identical requests in every slot must give identical tokens, run after run
(multi-stream prefills, batched ViT, batched decode graphs, fp8 paths).  python tools/soak.py [--fp8] [--iters N]"""
import argparse, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vision_inspection_system_amd.config import Qwen2VLConfig
from vision_inspection_system_amd.engine import Qwen2VLEngine
from vision_inspection_system_amd.weights import random_device_weights

ap = argparse.ArgumentParser()
ap.add_argument("--fp8", action="store_true")
ap.add_argument("--iters", type=int, default=4)
ap.add_argument("--batch", type=int, default=6)
ap.add_argument("--text-first", action="store_true", help="the reference's part order: common text prefix, shared per batch")
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = Qwen2VLConfig.qwen2_vl_7b()
eng = Qwen2VLEngine(cfg, random_device_weights(cfg, dev, 0), dev, max_ctx=4096, max_batch=a.batch,
                    decode_weights="fp8" if a.fp8 else "bf16", prefill_dtype="fp8" if a.fp8 else "bf16")
rng = np.random.default_rng(1)
frames = [torch.from_numpy(rng.integers(0, 256, (980, 980, 3), dtype=np.uint8)).to(dev) for _ in range(2)]
n_img = (980 // 14) ** 2 // 4
def ids_for(seed):
    r = np.random.default_rng(seed)
    if a.text_first:
        common = np.random.default_rng(1234).integers(0, 1000, 700).tolist()
        return common + [cfg.vision_start_id] + [cfg.image_token_id] * n_img + [cfg.vision_end_id] + r.integers(0, 1000, 5).tolist()
    return [cfg.vision_start_id] + [cfg.image_token_id] * n_img + [cfg.vision_end_id] + r.integers(0, 1000, 300).tolist()
reqs = [(ids_for(b % 2), [frames[b % 2]]) for b in range(a.batch)]      # two distinct requests, alternating over the slots
first = None
for it in range(a.iters):
    out = eng.generate_batch(reqs, max_new_tokens=24, ignore_eos=True)
    for b in range(a.batch):
        assert out[b] == out[b % 2], f"iteration {it}: slot {b} differs from slot {b % 2}"
    if first is None:
        first = out
    assert out == first, f"iteration {it}: result differs from the first iteration"
    single = eng.generate(reqs[0][0], reqs[0][1], max_new_tokens=24, ignore_eos=True)
    assert single[0] == out[0][0]
    print(f"iteration {it}: ok", out[0][:6], out[1][:6], flush=True)
print("SOAK OK")
