"""Race screen for the ping-pong 256x256 GEMM kernels (bf16 and fp8): integer-valued operands make every result exact,
so one stale / early LDS half-tile shows as a wrong integer.  Many launches at several shapes, with a bandwidth hog on
a second stream for part of them (memory load shifts the LDS-DMA landing times).  python tools/gemm_race_screen.py [iters]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vision_inspection_system_amd import hip

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda:0")
hip.load()
hog = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
side = torch.cuda.Stream()
shapes = [(2249, 12288, 3584), (2304, 12288, 1152), (4900, 5120, 1280), (2049, 12296, 2176), (8192, 8192, 1024),
          (2249, 3584, 3584), (4900, 1280, 1280), (2249, 1536, 3584), (1225, 5120, 5120)]   # the last four: 128 x 256 half-tiles
bad = 0
for (M, N, K) in shapes:
    g = torch.Generator(device="cpu").manual_seed(M + K)
    a = torch.randint(-2, 3, (M, K), generator=g).float()
    w = torch.randint(-1, 2, (N, K), generator=g).float()
    keep = torch.zeros(K)
    keep[torch.randperm(K, generator=g)[:120]] = 1.0
    w = w * keep
    ref = (a.to(dev) @ w.to(dev).t())
    assert float(ref.abs().max()) <= 256
    ab, wb = a.to(torch.bfloat16).to(dev), w.to(torch.bfloat16).to(dev)
    aq = a.to(torch.float8_e4m3fn).view(torch.uint8).to(dev)
    wq = w.to(torch.float8_e4m3fn).view(torch.uint8).to(dev)
    sa = torch.ones(M, dtype=torch.float32, device=dev)
    sw = torch.ones(N, dtype=torch.float32, device=dev)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    for it in range(iters):
        if it % 3 == 1:
            with torch.cuda.stream(side):
                hog.add_(1)
        hip.gemm(ab, wb, out=out)
        if not torch.equal(out.float(), ref):
            bad += 1
            print(f"bf16 MISMATCH {M}x{N}x{K} iteration {it}: {int((out.float() != ref).sum())} elements", flush=True)
        hip.gemm_fp8(aq, sa, wq, sw, out=out)
        if not torch.equal(out.float(), ref):
            bad += 1
            print(f"fp8 MISMATCH {M}x{N}x{K} iteration {it}: {int((out.float() != ref).sum())} elements", flush=True)
    torch.cuda.synchronize()
    print(f"{M}x{N}x{K}: {iters} x (bf16 + fp8) launches checked", flush=True)
print("RACE SCREEN", "FAILED" if bad else "OK")
sys.exit(1 if bad else 0)
