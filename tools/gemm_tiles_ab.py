"""Which tile kernel for the ragged production shapes, in the cache state they run in (operands and output not resident:
a 1 GiB fill between calls).  VIS_GEMM_TILE picks the kernel for the whole process (0/unset = the product's rule):
  for t in 0 7 6 5 4 1; do VIS_GEMM_TILE=$t python tools/gemm_tiles_ab.py; done"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vision_inspection_system_amd import hip
dev = torch.device("cuda:0")
hip.load()
flush = torch.empty(1 << 28, dtype=torch.float32, device=dev)
SHAPES = [("vit qkv", 4900, 3840, 1280, "b"), ("vit proj", 4900, 1280, 1280, "br"), ("vit fc1", 4900, 5120, 1280, "bq"),
          ("llm qkv", 2249, 4608, 3584, "b"), ("llm o", 2249, 3584, 3584, "r"), ("gate/up rest", 2249, 1536, 3584, "s"),
          ("gate/up all", 2249, 37888, 3584, "s")]
tile = os.environ.get("VIS_GEMM_TILE", "0")
for name, M, N, K, ep in SHAPES:
    a = torch.randn((M, K), device=dev).mul_(0.5).to(torch.bfloat16)
    w = (torch.randn((N, K), device=dev) / math.sqrt(K)).to(torch.bfloat16)
    bias = torch.randn((N,), device=dev).to(torch.bfloat16) if "b" in ep else None
    act = hip.ACT_SWIGLU if "s" in ep else (hip.ACT_QUICKGELU if "q" in ep else hip.ACT_NONE)
    out = torch.empty((M, N // 2 if act == hip.ACT_SWIGLU else N), dtype=torch.bfloat16, device=dev)
    res = out if "r" in ep else None
    if res is not None:
        out.normal_()
    ts = []
    for i in range(9):
        flush.fill_(float(i))
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); hip.gemm(a, w, bias=bias, residual=res, act=act, out=out); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    ts = sorted(ts[2:])
    t = ts[len(ts) // 2]
    print(f"tile={tile:>2s} {name:13s} {M}x{N}x{K}: {t:7.1f} us  {2.0 * M * N * K / t / 1e6:7.1f} TFLOP/s   (min {ts[0]:.1f})")
