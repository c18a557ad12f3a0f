"""Debug probe: streaming vs split form of the batched decode attention on structured V (V[key][d] = d / 16, then = key / 8)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vision_inspection_system_amd import hip
hip.load(); dev = torch.device("cuda:0")
def run(Hq, Hkv, B, T, ctxs, vmode):
    HD = 128
    g = torch.Generator().manual_seed(1)
    kc = torch.randn((B, Hkv, T, HD), generator=g).to(torch.bfloat16).to(dev)
    if vmode == "dim":
        vc = (torch.arange(HD).float() / 16).expand(B, Hkv, T, HD).contiguous().to(torch.bfloat16).to(dev)
    else:
        vc = (torch.arange(T).float()[:, None] / 8).expand(B, Hkv, T, HD).contiguous().to(torch.bfloat16).to(dev)
    ang = torch.rand((T, HD // 2), generator=g) * 6.28; emb = torch.cat((ang, ang), -1)
    cos_t = emb.cos().to(dev).unsqueeze(0).expand(B, -1, -1); sin_t = emb.sin().to(dev).unsqueeze(0).expand(B, -1, -1)
    qkv = torch.randn((B, (Hq + 2 * Hkv) * HD), generator=g).to(torch.bfloat16).to(dev)
    if vmode == "dim":
        qkv[:, (Hq + Hkv) * HD:] = (torch.arange(HD).float() / 16).repeat(Hkv).to(torch.bfloat16).to(dev)
    step = torch.tensor(ctxs, dtype=torch.int32, device=dev)
    ns = -(-T // hip.DECODE_KEYS_PER_SPLIT)
    po = torch.empty(B * Hq * ns * HD, dtype=torch.float32, device=dev); pml = torch.empty(B * Hq * ns * 2, dtype=torch.float32, device=dev)
    outs = {}
    for mode in ("2", "0"):
        os.environ["VIS_DECODE_ATTN_STREAM"] = mode
        k1, v1 = kc.clone(), vc.clone(); out = torch.zeros((B, Hq * HD), dtype=torch.bfloat16, device=dev)
        hip.decode_attn(qkv, cos_t, sin_t, k1, v1, step, po, pml, out, Hq, Hkv, HD, ns, HD ** -0.5)
        outs[mode] = out.float().cpu()
    d = (outs["2"] - outs["0"]).abs()
    print(vmode, Hq, Hkv, B, T, "ctx", ctxs, "max diff", float(d.max()))
    print("  stream head0 dims 0..23:", [round(float(x), 2) for x in outs["2"][0, :24]])
    print("  split  head0 dims 0..23:", [round(float(x), 2) for x in outs["0"][0, :24]])
for c in (0, 1, 3, 4, 7, 15, 16, 40):
    run(1, 1, 1, 256, [c], "dim")
run(1, 1, 1, 256, [15], "key")
