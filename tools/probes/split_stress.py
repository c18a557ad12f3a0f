"""Stress of the key-split attention merge (cross-workgroup, cross-XCD visibility without cache flushes): many launches
of the 4900 x 16 plan, part of them under a bandwidth hog on a second stream and with a second plan instance running on a
third stream, every output compared bit for bit with the first.  python tools/probes/split_stress.py [launches]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vision_inspection_system_amd import hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
dev = torch.device("cuda:0")
S, H, HD = 4900, 16, 80
g = torch.Generator(device="cpu").manual_seed(3)
q = torch.randn((H, S, HD), generator=g).to(torch.bfloat16).to(dev)
k = torch.randn((H, S, HD), generator=g).to(torch.bfloat16).to(dev)
ld = (S + 63) // 64 * 64
vt = torch.randn((H, HD, ld), generator=g).to(torch.bfloat16).to(dev)
plan = hip.make_vit_attn_plan([(0, S)], dev, H)
ref = torch.empty((S, H * HD), dtype=torch.bfloat16, device=dev)
hip.attn_prefill_plan(q, k, vt, ref, plan, HD ** -0.5)
torch.cuda.synchronize()
hog = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
side, third = torch.cuda.Stream(), torch.cuda.Stream()
out = torch.empty_like(ref)
out3 = torch.empty_like(ref)
bad = 0
for it in range(n):
    if it % 3 == 1:
        with torch.cuda.stream(side):
            hog.add_(1)
    if it % 5 == 2:
        with torch.cuda.stream(third):
            hip.attn_prefill_plan(q, k, vt, out3, plan, HD ** -0.5)
    hip.attn_prefill_plan(q, k, vt, out, plan, HD ** -0.5)
    if it % 50 == 49 or it == n - 1:
        torch.cuda.synchronize()
        if not torch.equal(out, ref) or (it >= 2 and not torch.equal(out3, ref)):
            bad += 1
            print(f"MISMATCH at launch {it}: {int((out != ref).sum())} / {int((out3 != ref).sum())} elements", flush=True)
    elif not (it % 7):
        torch.cuda.synchronize()
        if not torch.equal(out, ref):
            bad += 1
            print(f"MISMATCH at launch {it}: {int((out != ref).sum())} elements", flush=True)
counters = plan.ws[:plan.n_pairs * H * 4].view(torch.int32)
print(f"{n} launches, {bad} mismatching checks, counters all zero: {int(counters.abs().sum()) == 0}")
print("SPLIT STRESS OK" if bad == 0 else "SPLIT STRESS FAILED")
