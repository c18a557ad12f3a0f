"""Does a GEMV run faster when its weights were read just before (L2 / Infinity Cache residency)?
cold: 1 GiB flush read, then the GEMV;  pre: flush, a plain read of W (torch sum), then the GEMV;  self: GEMV twice."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vision_inspection_system_amd import hip
dev = torch.device("cuda:0")
hip.load()
flush = torch.zeros(1024 * 1024 * 1024 // 4, device=dev)
def med(f, pre, n=15):
    ts = []
    for _ in range(n):
        pre()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); f(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    return sorted(ts)[n // 2]
for name, (N, K, sw) in {"qkv": (4608, 3584, False), "o": (3584, 3584, False), "gateup": (37888, 3584, True),
                         "down": (3584, 18944, False)}.items():
    w = (torch.randn((N, K), device=dev) / math.sqrt(K)).to(torch.bfloat16)
    x = torch.randn((K,), device=dev).to(torch.bfloat16)
    out = torch.empty((N // 2 if sw else N,), dtype=torch.bfloat16, device=dev)
    act = hip.ACT_SWIGLU if sw else hip.ACT_NONE
    hip.gemv(x, w, out, act=act); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        hip.gemv(x, w, out, act=act)
    wi = w.view(torch.int32)
    cold = med(g.replay, lambda: flush.sum())
    pre = med(g.replay, lambda: (flush.sum(), wi.sum()))
    slf = med(g.replay, lambda: (flush.sum(), g.replay()))
    empty = med(lambda: None, lambda: None)
    print(f"{name:7s} {N*K*2/1e6:7.1f} MB   cold {cold:6.1f} us   after a plain read of W {pre:6.1f} us   after itself {slf:6.1f} us   (empty event pair {empty:4.1f} us)")
    del w
