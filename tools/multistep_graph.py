"""Experiment: one hipGraph per token vs one graph holding 4 decode steps (graph-boundary cost)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from vision_inspection_system_amd.config import Qwen2VLConfig
from vision_inspection_system_amd.engine import Qwen2VLEngine
from vision_inspection_system_amd.weights import random_device_weights
dev = torch.device("cuda:0")
cfg = Qwen2VLConfig.qwen2_vl_7b()
eng = Qwen2VLEngine(cfg, random_device_weights(cfg, dev, 0), dev, max_ctx=4096)
rng = np.random.default_rng(0)
frame = torch.from_numpy(rng.integers(0, 256, (980, 980, 3), dtype=np.uint8)).to(dev)
ids = [cfg.vision_start_id] + [cfg.image_token_id] * 1225 + [cfg.vision_end_id] + rng.integers(0, 1000, 1022).tolist()
def run(k):
    eng.prefill(ids, [frame], max_new_tokens=200)
    eng._decode_step(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(k):
            eng._decode_step()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(120 // k):
        g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / 120
for k in (1, 4, 1, 4, 8):
    print(k, "steps per graph:", round(run(k), 4), "ms/token", flush=True)
