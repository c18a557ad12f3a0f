"""Does a hipGraph shorten the prompt pass?  ~510 dependent kernels per image: eager stream launches vs one captured graph
(same kernels, same order).  python tools/prefill_graph_probe.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vision_inspection_system_amd.config import Qwen2VLConfig
from vision_inspection_system_amd.engine import Qwen2VLEngine
from vision_inspection_system_amd.weights import random_device_weights
dev = torch.device("cuda:0")
cfg = Qwen2VLConfig.qwen2_vl_7b()
eng = Qwen2VLEngine(cfg, random_device_weights(cfg, dev, 0), dev, max_ctx=4096, max_batch=1)
rng = np.random.default_rng(7)
frame = torch.from_numpy(rng.integers(0, 256, (980, 980, 3), dtype=np.uint8)).to(dev)
n_img = (980 // 14) ** 2 // 4
ids = rng.integers(0, 1000, 1000).tolist() + [cfg.vision_start_id] + [cfg.image_token_id] * n_img + [cfg.vision_end_id] + rng.integers(0, 1000, 22).tolist()
ids_dev = torch.tensor(ids, dtype=torch.int32, device=dev)

def timed(fn, n=5):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

eager = lambda: eng.prefill(ids, [frame], ids_dev=ids_dev, max_new_tokens=128)
for _ in range(3):
    eager()
print(f"eager  : {timed(eager):.2f} ms per prompt pass (S = {len(ids)})")
first = int(eng.tokens_b[0][len(ids)]) if eng.tokens_b.dim() > 1 else None
logits_e = eng.logits_b[0].clone()
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    eager()
    with torch.cuda.graph(g, stream=side):
        eager()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
for _ in range(3):
    g.replay()
print(f"graph  : {timed(g.replay):.2f} ms per prompt pass")
print("logits identical:", bool(torch.equal(logits_e, eng.logits_b[0])))
print(f"eager again: {timed(eager):.2f} ms")
