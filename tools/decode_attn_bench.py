"""Decode attention + combine at batch B (exact 7B geometry: 28 query / 4 KV heads, d = 128, context ctx):
python tools/decode_attn_bench.py [B] [ctx]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vision_inspection_system_amd import hip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ctx = int(sys.argv[2]) if len(sys.argv) > 2 else 2300
dev = torch.device("cuda:0")
hip.load()
Hq, Hkv, HD, T = 28, 4, 128, (int(sys.argv[3]) if len(sys.argv) > 3 else 4096)
g = torch.Generator(device="cpu").manual_seed(0)
kc = torch.randn((B, Hkv, T, HD), generator=g).to(torch.bfloat16).to(dev)
vc = torch.randn((B, Hkv, T, HD), generator=g).to(torch.bfloat16).to(dev)
ang = torch.rand((T, HD // 2), generator=g) * 6.28
emb = torch.cat((ang, ang), -1)
cos_t = emb.cos().to(dev).repeat(B, 1, 1).contiguous()
sin_t = emb.sin().to(dev).repeat(B, 1, 1).contiguous()
qkv = torch.randn((B, (Hq + 2 * Hkv) * HD), generator=g).to(torch.bfloat16).to(dev)
step = torch.full((B,), ctx, dtype=torch.int32, device=dev)
nsplit = T // hip.DECODE_KEYS_PER_SPLIT
part_o = torch.empty(B * Hq * nsplit * HD, dtype=torch.float32, device=dev)
part_ml = torch.empty(B * Hq * nsplit * 2, dtype=torch.float32, device=dev)
out = torch.empty((B, Hq * HD), dtype=torch.bfloat16, device=dev)
flush = torch.zeros(512 * 1024 * 1024 // 4, device=dev)
run = lambda: hip.decode_attn(qkv, cos_t, sin_t, kc, vc, step, part_o, part_ml, out, Hq, Hkv, HD, nsplit, HD ** -0.5)
run(); torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    run()
ts = []
for _ in range(9):
    flush.sum()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); gr.replay(); e.record(); torch.cuda.synchronize()
    ts.append(s.elapsed_time(e) * 1e3)
t = sorted(ts)[4]
kv_bytes = B * Hkv * (ctx + 1) * HD * 2 * 2
print(f"B={B} ctx={ctx}: attention + combine {t:.1f} us (incl. ~8 us of graph launch / event overhead), KV {kv_bytes/1e6:.0f} MB -> {kv_bytes/t/1e6:.2f} TB/s")
