"""Attention-only microbench (ViT varlen d=80 and LLM causal d=128) for profiling runs."""
import sys, os, argparse
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vision_inspection_system_amd import hip

ap = argparse.ArgumentParser()
ap.add_argument("--which", default="vit")
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--plan", type=int, default=1, help="0: plain 128-row items; 1: planner (full + half items)")
a = ap.parse_args()
dev = torch.device("cuda:0")
hip.load()
S, Hq, Hkv, HD, causal = (4900, 16, 16, 80, False) if a.which == "vit" else (2249, 28, 4, 128, True)
q = torch.randn((Hq, S, HD), device=dev).to(torch.bfloat16)
k = torch.randn((Hkv, S, HD), device=dev).to(torch.bfloat16)
ld = (S + 63) // 64 * 64
vt = torch.randn((Hkv, HD, ld), device=dev).to(torch.bfloat16)
o = torch.empty((S, Hq * HD), dtype=torch.bfloat16, device=dev)
work = hip.make_attn_work([(0, S)], causal, dev, heads=Hq if a.plan else 0)
if os.environ.get("VIS_ATTN_FULLS") is not None and not causal:   # experiment: k full items per head, rest as 64-row items
    nf = int(os.environ["VIS_ATTN_FULLS"])
    items = [(q0, 128, 0, S) for q0 in range(0, nf * 128, 128)] + [(q0, min(64, S - q0), 0, S) for q0 in range(nf * 128, S, 64)]
    work = torch.tensor(items, dtype=torch.int32, device=dev).reshape(-1, 4).contiguous()
if os.environ.get("VIS_ATTN_FAKE_SPLIT") is not None and not causal:   # TIMING experiment (results of the split rows are wrong):
    nf = int(os.environ["VIS_ATTN_FAKE_SPLIT"])                           # nf full items per head, the rest as two half-key items
    order = os.environ.get("VIS_ATTN_SPLIT_ORDER", "last")
    half = ((S // 2) + 63) // 64 * 64
    full = [(q0, min(128, S - q0), 0, S) for q0 in range(0, min(S, nf * 128), 128)]
    split = [(q0, min(128, S - q0), a0, a1) for q0 in range(nf * 128, S, 128) for (a0, a1) in ((0, half), (half, S))]
    items = full + split if order == "last" else split + full
    work = torch.tensor(items, dtype=torch.int32, device=dev).reshape(-1, 4).contiguous()
if os.environ.get("VIS_ATTN_CAUSAL") is not None and causal:   # experiment: causal item orders / block sizes
    mode, bq = (int(x) for x in os.environ["VIS_ATTN_CAUSAL"].split(","))
    items = [(q0, min(bq, S - q0), 0, S) for q0 in range(0, S, bq)]
    items.sort(key=lambda it: -(it[0] + it[1]))
    half = (len(items) + 1) // 2
    if mode == 1:      # heavy half descending, light half ascending (lightest first)
        items = items[:half] + items[half:][::-1]
    elif mode == 2:    # alternate heavy / light
        hv, lt = items[:half], items[half:][::-1]
        items = [x for pair in zip(hv, lt + [None] * (len(hv) - len(lt))) for x in pair if x is not None]
    work = torch.tensor(items, dtype=torch.int32, device=dev).reshape(-1, 4).contiguous()
print('items', work.shape[0], 'half', int((work[:, 1] <= 64).sum()))
run = lambda: hip.attn_prefill(q, k, vt, o, work, causal, HD ** -0.5)
if (not causal and a.plan and os.environ.get("VIS_ATTN_SPLIT", "1") != "0" and os.environ.get("VIS_ATTN_FULLS") is None
        and os.environ.get("VIS_ATTN_FAKE_SPLIT") is None):
    vplan = hip.make_vit_attn_plan([(0, S)], dev, Hq)        # what the engine launches: key-split items merged in the kernel
    print('key-split plan: items', vplan.work.shape[0], 'pairs', vplan.n_pairs)
    run = lambda: hip.attn_prefill_plan(q, k, vt, o, vplan, HD ** -0.5)
for _ in range(2):
    run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(a.reps):
    run()
e.record()
torch.cuda.synchronize()
t = s.elapsed_time(e) / a.reps * 1e-3
fl = 4.0 * S * S * HD * Hq * (0.5 if causal else 1.0)
print(f"{a.which}: {t*1e3:.3f} ms  {fl/t/1e12:.1f} TFLOP/s")
if causal and HD == 128:
    pw = hip.make_attn_pairs(0, S, dev)
    for _ in range(2):
        hip.attn_prefill_pairs(q, k, vt, o, pw, HD ** -0.5)
    torch.cuda.synchronize()
    s.record()
    for _ in range(a.reps):
        hip.attn_prefill_pairs(q, k, vt, o, pw, HD ** -0.5)
    e.record()
    torch.cuda.synchronize()
    t = s.elapsed_time(e) / a.reps * 1e-3
    print(f"{a.which} paired blocks ({pw.shape[0]} pairs): {t*1e3:.3f} ms  {fl/t/1e12:.1f} TFLOP/s")
