"""ViT attention 4900 x 16 heads: the key-split plan (vis_attn_prefill_split) against the plain planner items."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vision_inspection_system_amd import hip
dev = torch.device("cuda:0")
S, H, HD = 4900, 16, 80
q = torch.randn((H, S, HD), device=dev).to(torch.bfloat16)
k = torch.randn((H, S, HD), device=dev).to(torch.bfloat16)
ld = (S + 63) // 64 * 64
vt = torch.randn((H, HD, ld), device=dev).to(torch.bfloat16)
o = torch.empty((S, H * HD), dtype=torch.bfloat16, device=dev)
def t(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
plain = hip.make_vit_attn_plan([(0, S)], dev, H, split=False)
split = hip.make_vit_attn_plan([(0, S)], dev, H, split=True)
for r in range(3):
    print(f"plain {t(lambda: hip.attn_prefill_plan(q, k, vt, o, plain, HD ** -0.5)):.1f} us   "
          f"split {t(lambda: hip.attn_prefill_plan(q, k, vt, o, split, HD ** -0.5)):.1f} us  (items {split.work.shape[0]}, pairs {split.n_pairs})")
