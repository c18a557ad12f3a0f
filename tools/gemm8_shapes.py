"""fp8 GEMM time per production shape under the current VIS_GEMM8_TILE setting (unset = the library's cost model, 1 = 128 x 128,
4 = 256 x 256 ping-pong): run once per setting and compare the columns - the check of vis_gemm_fp8's tile choice.
  for t in "" 1 4; do VIS_GEMM8_TILE=$t python tools/gemm8_shapes.py; done"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vision_inspection_system_amd import hip
dev = torch.device("cuda:0")
hip.load()
SHAPES = [  # (name, M, N, K, act)
    ("vit qkv   x4", 19600, 3840, 1280, 0), ("vit proj  x4", 19600, 1280, 1280, 0), ("vit fc1   x4", 19600, 5120, 1280, 1),
    ("vit fc2   x4", 19600, 1280, 5120, 0), ("llm qkv   x4", 5156, 4608, 3584, 0), ("llm o     x4", 5156, 3584, 3584, 0),
    ("llm gate/up x4", 5156, 37888, 3584, 3), ("llm qkv   x1", 2249, 4608, 3584, 0), ("llm o     x1", 2249, 3584, 3584, 0),
    ("llm gate/up x1", 2249, 37888, 3584, 3), ("vit qkv   x1", 4900, 3840, 1280, 0), ("vit fc1   x1", 4900, 5120, 1280, 1),
    ("vit fc2   x1", 4900, 1280, 5120, 0), ("merger fc1 x4", 4900, 5120, 5120, 2), ("merger fc2 x4", 4900, 3584, 5120, 0),
]
tag = os.environ.get("VIS_GEMM8_TILE", "") or "model"
for name, M, N, K, act in SHAPES:
    a = torch.randn((M, K), device=dev).to(torch.bfloat16)
    w = (torch.randn((N, K), device=dev) / math.sqrt(K)).to(torch.bfloat16)
    aq, sa = hip.quant_rows_fp8(a)
    wq, sw = hip.quantize_fp8_rows(w)
    bias = torch.randn((N,), device=dev).to(torch.bfloat16) if act in (1, 2) else None
    out = torch.empty((M, N // 2 if act == 3 else N), dtype=torch.bfloat16, device=dev)
    run = lambda: hip.gemm_fp8(aq, sa, wq, sw, bias=bias, act=act, out=out)
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
    print(f"tile={tag:5s} {name:16s} {M:6d} x {N:6d} x {K:5d}: {t:8.1f} us  {2.0*M*N*K/t/1e6:7.1f} TFLOP/s", flush=True)
