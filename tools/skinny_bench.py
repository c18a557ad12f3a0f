"""Batched-decode projection microbench (gemm_decode + finalize), graph-replayed: host launch cost excluded."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vision_inspection_system_amd import hip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
hip.load()
flush = torch.zeros(1024 * 1024 * 1024 // 4, device=dev)  # read-only flush: leaves CLEAN lines, like the
# previous projection's weights do in the real decode loop (a writing flush makes the kernel compete with write-backs)
shapes = {"qkv": (4608, 3584, False), "o": (3584, 3584, False), "gateup": (37888, 3584, True),
          "down": (3584, 18944, False), "lm_head": (152064, 3584, False)}
for name, (N, K, sw) in shapes.items():
    w = (torch.randn((N, K), device=dev) / math.sqrt(K)).to(torch.bfloat16)
    x = torch.randn((B, K), device=dev).to(torch.bfloat16)
    part = torch.empty(16 * 16 * N, dtype=torch.float32, device=dev)
    if name == "lm_head":
        out = torch.empty((B, N), dtype=torch.float32, device=dev)
        run = lambda: hip.decode_gemm(x, w, out=out)
    else:
        out = torch.empty((B, N // 2 if sw else N), dtype=torch.bfloat16, device=dev)
        def run():
            ks = hip.decode_gemm(x, w, part=part)
            hip.skinny_finalize(part, ks, out, N, swiglu=sw)
    run(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run()
    ts = []
    for _ in range(5):
        flush.sum()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); g.replay(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e-3)
    t = sorted(ts)[2]
    print(f"decode-gemm B={B} {name:8s} ks={hip.load().vis_gemm_decode_ksplit(N, K):2d} {t*1e6:8.1f} us  {N*K*2/t/1e9:8.1f} GB/s")
    del w
