"""Batched-decode stream kernel alone (no finalisation), graph-replayed, cold L2: python tools/probes/dec_depth_bench.py B
Used with probe builds of csrc/libvis_hip.so (-DGEMM3W_DEPTH=5) to see what the ring depth is worth at 33..64 rows."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vision_inspection_system_amd import hip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
hip.load()
flush = torch.zeros(1024 * 1024 * 1024 // 4, device=dev)
shapes = {"qkv": (4608, 3584), "o": (3584, 3584), "gateup": (37888, 3584), "down": (3584, 18944), "lm_head": (152064, 3584)}
for name, (N, K) in shapes.items():
    w = (torch.randn((N, K), device=dev) / math.sqrt(K)).to(torch.bfloat16)
    x = torch.randn((B, K), device=dev).to(torch.bfloat16)
    ks = hip.load().vis_gemm_decode_ksplit(N, K)
    part = torch.empty(ks * hip.part_rows(B) * N, dtype=torch.float32, device=dev)
    run = lambda: hip.decode_gemm(x, w, part=part)
    run(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run()
    ts = []
    for _ in range(7):
        flush.sum()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); g.replay(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e-3)
    t = sorted(ts)[3]
    print(f"B={B} {name:8s} ks={ks:2d} {t*1e6:8.1f} us  {N*K*2/t/1e9:8.1f} GB/s")
    del w
