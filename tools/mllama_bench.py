"""Row f2 timing at exact Llama-3.2-11B-Vision shapes (seeded random weights): one 1024x1024 image (2x2 tiles ->
6404 vision tokens), a prompt of --prompt-tokens text tokens followed by the image token (the reference's part
order), --new-tokens greedy tokens.  Prints one JSON line.  Not the headline bench (that is bench.py, Qwen2-VL-7B)."""
import argparse, json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vision_inspection_system_amd.mllama_engine import MllamaEngine
from vision_inspection_system_amd.mllama_weights import MllamaConfig, random_device_weights

ap = argparse.ArgumentParser()
ap.add_argument("--prompt-tokens", type=int, default=700)
ap.add_argument("--new-tokens", type=int, default=128)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--model", default="11b", choices=["11b", "tiny"])
ap.add_argument("--batch", type=int, default=1, help="> 1: that many images per step, per-image prompt pass + ONE shared decode loop")
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = MllamaConfig.mllama_11b() if a.model == "11b" else MllamaConfig.tiny()
eng = MllamaEngine(cfg, random_device_weights(cfg, dev, 0), dev, max_ctx=2048 if a.model == "11b" else 1024, max_batch=a.batch)
rng = np.random.default_rng(0)
side = 1024 if a.model == "11b" else 100
frame = torch.from_numpy(rng.integers(0, 256, (side, side, 3), dtype=np.uint8)).to(dev)
ids = [1] + rng.integers(1000 if a.model == "11b" else 3, cfg.vocab - 8, a.prompt_tokens).tolist() + [cfg.image_token_id, 5, 6]
res = []
for it in range(a.steps + 1):
    s, m, e = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    s.record()
    if a.batch > 1:      # the product path of verify_many: stacked tower per group, two-stream prompt passes, one decode loop
        eng.generate_batch([(ids, frame)] * a.batch, max_new_tokens=a.new_tokens, stop_on_eos=False)
        m = None
    else:
        eng.prefill(ids, frame)
        m.record()
        eng.decode(a.new_tokens - 1, use_graph=True)
    e.record()
    torch.cuda.synchronize()
    if it:
        res.append((s.elapsed_time(m), m.elapsed_time(e)) if m is not None else
                   (eng.last_timing["prefill_ms"], eng.last_timing["decode_ms"]))
pre = float(np.mean([r[0] for r in res])); dec = float(np.mean([r[1] for r in res]))
wbytes = sum(t.numel() * 2 for lw in eng.w.layers for t in (lw.qkv_w, lw.o_w, lw.gateup_w, lw.down_w) if t is not None) + eng.w.lm_head.numel() * 2
print(json.dumps({"model": cfg.name, "prompt_tokens": len(ids), "vision_tokens": eng.TP, "new_tokens": a.new_tokens,
                  "prefill_ms": pre, "decode_ms": dec, "ms_per_token": dec / (a.new_tokens - 1),
                  "batch": a.batch, "images_per_s": 1000.0 * a.batch / (pre + dec), "prefill_ms_per_image": pre / a.batch,
                  "decode_weight_GBps": wbytes / (dec / (a.new_tokens - 1) * 1e-3) / 1e9}))
