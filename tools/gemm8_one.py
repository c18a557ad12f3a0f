"""One fp8 GEMM shape, a few launches - the target of the PMC passes in tools/fp8_gemm_pmc.sh: python tools/gemm8_one.py M N K ACT [reps]"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vision_inspection_system_amd import hip
M, N, K, act = (int(x) for x in sys.argv[1:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
dev = torch.device("cuda:0")
hip.load()
a = torch.randn((M, K), device=dev).to(torch.bfloat16)
w = (torch.randn((N, K), device=dev) / math.sqrt(K)).to(torch.bfloat16)
aq, sa = hip.quant_rows_fp8(a)
wq, sw = hip.quantize_fp8_rows(w)
bias = torch.randn((N,), device=dev).to(torch.bfloat16) if act in (1, 2) else None
out = torch.empty((M, N // 2 if act == 3 else N), dtype=torch.bfloat16, device=dev)
for _ in range(reps):
    hip.gemm_fp8(aq, sa, wq, sw, bias=bias, act=act, out=out)
torch.cuda.synchronize()
