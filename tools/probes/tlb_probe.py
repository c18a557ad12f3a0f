"""Is the 'GEMV right after itself is faster' effect (tools/prefetch_probe.py: gate/up 59.4 -> 50.5 us) address translation?
cold: 1 GiB flush read, then the GEMV;  touch: flush, then ONE 16-byte read per `gran` bytes of W issued from every XCD
(vis_gather_rows over a page-strided view, eight rotations so that every page is touched by workgroups of all eight XCDs),
then the GEMV;  self: flush, GEMV, GEMV.   python tools/probes/tlb_probe.py"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vision_inspection_system_amd import hip
dev = torch.device("cuda:0")
hip.load()
flush = torch.zeros(1024 * 1024 * 1024 // 4, device=dev)


def med(f, pre, n=11):
    ts = []
    for _ in range(n):
        pre()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); f(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    return sorted(ts)[n // 2]


for name, (N, K, sw) in {"o": (3584, 3584, False), "gateup": (37888, 3584, True), "down": (3584, 18944, False)}.items():
    w = (torch.randn((N, K), device=dev) / math.sqrt(K)).to(torch.bfloat16)
    x = torch.randn((K,), device=dev).to(torch.bfloat16)
    out = torch.empty((N // 2 if sw else N,), dtype=torch.bfloat16, device=dev)
    act = hip.ACT_SWIGLU if sw else hip.ACT_NONE
    hip.gemv(x, w, out, act=act); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        hip.gemv(x, w, out, act=act)
    cold = med(g.replay, lambda: flush.sum())
    slf = med(g.replay, lambda: (flush.sum(), g.replay()))
    line = f"{name:7s} {N * K * 2 / 1e6:7.1f} MB   cold {cold:6.1f} us   after itself {slf:6.1f} us"
    for gran in (4096, 65536, 2 << 20):
        el = gran // 2
        n_pages = (N * K) // el
        if n_pages < 8:
            continue
        n16 = (N * K) // 8                                    # the weights as rows of 16 bytes
        sink = torch.empty((n_pages, 8), dtype=torch.bfloat16, device=dev)
        ids = [(((torch.arange(n_pages, device=dev, dtype=torch.int64) + 4 * k) % n_pages) * (el // 8)).to(torch.int32).contiguous()
               for k in range(8)]

        def touch():
            flush.sum()
            for k in range(8):
                rc = hip.load().vis_gather_rows(w.data_ptr(), ids[k].data_ptr(), sink.data_ptr(), n_pages, 8, n16,
                                                torch.cuda.current_stream().cuda_stream)
                assert rc == 0
        # vis_gather_rows reads D = 8 elements (16 bytes) of row ids[i] of a table whose rows are `gran` bytes apart
        t = med(g.replay, touch)
        line += f"   touch/{gran >> 10}K {t:6.1f}"
    print(line, flush=True)
    del w
