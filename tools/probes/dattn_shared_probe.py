"""EXPERIMENT (timing only): the 64-sequence streaming decode attention when the sequences' caches overlap in memory
(batch stride of 16 rows instead of a whole cache: they read almost the same lines) - an upper bound for reading a
shared text prefix's K / V from ONE copy instead of from every slot's copy."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vision_inspection_system_amd import hip
B, ctx, T = 64, 2300, 4096
dev = torch.device("cuda:0")
hip.load()
Hq, Hkv, HD = 28, 4, 128
g = torch.Generator(device="cpu").manual_seed(0)
base_k = torch.randn((B * Hkv * T * HD + B * 16 * HD,), generator=g).to(torch.bfloat16).to(dev)
base_v = torch.randn((B * Hkv * T * HD + B * 16 * HD,), generator=g).to(torch.bfloat16).to(dev)
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
for name, bs in (("own cache per sequence", Hkv * T * HD), ("overlapping caches (stride 16 rows)", 16 * HD)):
    kc = base_k.as_strided((B, Hkv, T, HD), (bs, T * HD, HD, 1))
    vc = base_v.as_strided((B, Hkv, T, HD), (bs, T * HD, HD, 1))
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
    print(f"{name}: {sorted(ts)[4]:.1f} us")
