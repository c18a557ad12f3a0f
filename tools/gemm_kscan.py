"""Fixed cost per tile vs cost per K-step of the GEMM tile kernels: one round of the chip (4096 x 4096 = 256 tiles of
256 x 256) at K = 256 .. 8192, for each forced tile kernel; prints t(K) and the least-squares a + b * (K / 64).
python tools/gemm_kscan.py  (set VIS_GEMM_TILE in the environment to pick the kernel: 7 = ping-pong 256x256;
KS_FP8=1: the fp8 kernels instead, VIS_GEMM8_TILE=4 / 1 forces the 256x256 / 128x128 tile; K-steps are then 128 wide)"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vision_inspection_system_amd import hip
dev = torch.device("cuda:0")
hip.load()
M = int(os.environ.get("KS_M", "4096")); N = int(os.environ.get("KS_N", "4096"))
act = int(os.environ.get("KS_ACT", "0"))
FP8 = os.environ.get("KS_FP8") == "1"
rows = []
for K in (256, 512, 1024, 1280, 2048, 3584, 4096, 8192):
    a = torch.randn((M, K), device=dev).to(torch.bfloat16)
    w = (torch.randn((N, K), device=dev) / math.sqrt(K)).to(torch.bfloat16)
    bias = torch.randn((N,), device=dev).to(torch.bfloat16) if act in (1, 2) else None
    out = torch.empty((M, N // 2 if act == 3 else N), dtype=torch.bfloat16, device=dev)
    if FP8:
        aq, sa = hip.quant_rows_fp8(a)
        wq, sw = hip.quantize_fp8_rows(w)
        run = lambda: hip.gemm_fp8(aq, sa, wq, sw, bias=bias, act=act, out=out)
    else:
        run = lambda: hip.gemm(a, w, bias=bias, act=act, out=out)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(4):
            run()
        e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / 4 * 1e3)
    ts.sort()
    t = ts[len(ts) // 2]
    rows.append((K, t))
    print(f"{'fp8 ' if FP8 else ''}tile={os.environ.get('VIS_GEMM8_TILE' if FP8 else 'VIS_GEMM_TILE','auto')} M={M} N={N} K={K:5d}: {t:8.1f} us  {2.0*M*N*K/t/1e6:7.1f} TFLOP/s")
xs = [k / 64 for k, _ in rows]; ys = [t for _, t in rows]
n = len(xs); sx, sy = sum(xs), sum(ys); sxx = sum(x * x for x in xs); sxy = sum(x * y for x, y in zip(xs, ys))
b = (n * sxy - sx * sy) / (n * sxx - sx * sx); a = (sy - b * sx) / n
print(f"fit: t = {a:.1f} us + {b:.3f} us per 64-wide K-step")
