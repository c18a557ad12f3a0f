"""Skinny (batched-decode) GEMM microbench, graph-replayed so host launch cost is excluded."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vision_inspection_system_amd import hip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
hip.load()
flush = torch.zeros(512 * 1024 * 1024 // 4, device=dev)
shapes = {"qkv": (4608, 3584, 0), "o": (3584, 3584, 0), "gateup": (37888, 3584, 3), "down": (3584, 18944, 0), "lm_head": (152064, 3584, 0)}
for name, (N, K, act) in shapes.items():
    w = (torch.randn((N, K), device=dev) / math.sqrt(K)).to(torch.bfloat16)
    x = torch.randn((B, K), device=dev).to(torch.bfloat16)
    out = torch.empty((B, N // 2 if act else N), dtype=torch.float32 if name == "lm_head" else torch.bfloat16, device=dev)
    part = torch.empty(16 * 16 * N, dtype=torch.float32, device=dev) if (act == 0 and name != "lm_head") else None
    nw = torch.randn(K, device=dev).to(torch.bfloat16)
    rstd = torch.ones(16, device=dev)
    def run():
        hip.skinny_gemm(x, w, out, part=part, norm_w=nw, rstd=rstd, act=act)
    run(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run()
    ts = []
    for _ in range(5):
        flush.add_(1.0)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); g.replay(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e-3)
    t = sorted(ts)[2]
    print(f"skinny B={B} {name:8s} nt={os.environ.get('VIS_SKINNY_NT','1')} {t*1e6:8.1f} us  {N*K*2/t/1e9:8.1f} GB/s")
    del w
