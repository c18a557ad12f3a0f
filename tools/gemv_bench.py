"""GEMV microbench (bf16 vs fp8 weights), graph-replayed with a read-only cache flush between replays."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vision_inspection_system_amd import hip
dev = torch.device("cuda:0")
hip.load()
flush = torch.zeros(1024 * 1024 * 1024 // 4, device=dev)
for name, (N, K, sw) in {"qkv": (4608, 3584, False), "o": (3584, 3584, False), "gateup": (37888, 3584, True),
                         "down": (3584, 18944, False), "lm_head": (152064, 3584, False)}.items():
    w = (torch.randn((N, K), device=dev) / math.sqrt(K)).to(torch.bfloat16)
    x = torch.randn((K,), device=dev).to(torch.bfloat16)
    wq, sc = hip.quantize_fp8_rows(w)
    out = torch.empty((N // 2 if sw else N,), dtype=torch.bfloat16, device=dev)
    act = hip.ACT_SWIGLU if sw else hip.ACT_NONE
    for kind, run, nbytes in (("bf16", lambda: hip.gemv(x, w, out, act=act), N * K * 2),
                              ("fp8 ", lambda: hip.gemv_fp8(x, wq, sc, out, act=act), N * K)):
        run(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(4):
                run()
        ts = []
        for _ in range(5):
            flush.sum()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); g.replay(); e.record(); torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) * 1e-3 / 4)
        t = sorted(ts)[2]
        print(f"gemv {kind} {name:8s} {t*1e6:8.1f} us  {nbytes/t/1e9:8.1f} GB/s")
    del w, wq
