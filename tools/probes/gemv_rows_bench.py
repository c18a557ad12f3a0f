"""Multi-row GEMV (vis_gemv_*_rows) per projection shape and row count, graph-replayed, cold L2."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vision_inspection_system_amd import hip
dev = torch.device("cuda:0")
hip.load()
flush = torch.zeros(1024 * 1024 * 1024 // 4, device=dev)
shapes = {"qkv": (4608, 3584, 0), "o": (3584, 3584, 0), "gateup": (37888, 3584, 3), "down": (3584, 18944, 0), "lm_head": (152064, 3584, 0)}
for fp8 in (False, True):
    for name, (N, K, act) in shapes.items():
        w = (torch.randn((N, K), device=dev) / math.sqrt(K)).to(torch.bfloat16)
        wq, sw = hip.quantize_fp8_rows(w) if fp8 else (None, None)
        line = f"{'fp8 ' if fp8 else 'bf16'} {name:8s}"
        for B in (1, 2, 4):
            x = torch.randn((B, K), device=dev).to(torch.bfloat16)
            out = torch.empty((B, N // 2 if act else N), dtype=torch.bfloat16, device=dev)
            run = (lambda: hip.gemv_fp8_rows(x, wq, sw, out, act=act)) if fp8 else (lambda: hip.gemv_rows(x, w, out, act=act))
            run(); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                run()
            ts = []
            for _ in range(7):
                flush.sum()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(); g.replay(); e.record(); torch.cuda.synchronize()
                ts.append(s.elapsed_time(e) * 1e3)
            t = sorted(ts)[3]
            line += f"   B={B}: {t:7.1f} us {N * K * (1 if fp8 else 2) / t / 1e6:6.2f} TB/s"
        print(line, flush=True)
        del w
