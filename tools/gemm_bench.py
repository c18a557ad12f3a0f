"""GEMM-only microbench for profiling runs: python tools/gemm_bench.py M N K [reps]."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vision_inspection_system_amd import hip
M, N, K = (int(x) for x in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dev = torch.device("cuda:0")
hip.load()
a = (torch.randn((M, K), device=dev)).to(torch.bfloat16)
w = (torch.randn((N, K), device=dev) / math.sqrt(K)).to(torch.bfloat16)
out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
for _ in range(2):
    hip.gemm(a, w, out=out)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(reps):
    hip.gemm(a, w, out=out)
e.record()
torch.cuda.synchronize()
t = s.elapsed_time(e) / reps * 1e-3
ref = a.float() @ w.float().t()
err = (out.float() - ref).abs().max().item()
print(f"max abs err vs fp32 matmul: {err:.4f} (ref max {ref.abs().max().item():.2f})")
assert os.environ.get("VIS_NOCHECK") or err < 0.06 * max(1.0, ref.abs().max().item()), "GEMM mismatch"
if os.environ.get("VIS_FP8"):
    aq, sa = hip.quant_rows_fp8(a)
    wq, sw = hip.quantize_fp8_rows(w)
    o8 = torch.empty_like(out)
    for _ in range(2):
        hip.gemm_fp8(aq, sa, wq, sw, out=o8)
    torch.cuda.synchronize()
    s.record()
    for _ in range(reps):
        hip.gemm_fp8(aq, sa, wq, sw, out=o8)
    e.record(); torch.cuda.synchronize()
    t8 = s.elapsed_time(e) / reps * 1e-3
    s.record()
    for _ in range(reps):
        hip.quant_rows_fp8(a, aq, sa)
    e.record(); torch.cuda.synchronize()
    tq = s.elapsed_time(e) / reps * 1e-3
    print(f"fp8 MFMA: {t8*1e3:.3f} ms {2.0*M*N*K/t8/1e12:.1f} TFLOP/s  (quantise A: {tq*1e6:.1f} us)  max diff vs bf16 {(o8.float()-out.float()).abs().max().item():.3f}")
if os.environ.get("VIS_SPLITK"):
    ks = int(os.environ["VIS_SPLITK"])
    work = torch.empty(ks * M * N, dtype=torch.float32, device=dev)
    out2 = torch.empty_like(out)
    for _ in range(2):
        hip.gemm_splitk(a, w, work, ks, out=out2)
    torch.cuda.synchronize()
    s.record()
    for _ in range(reps):
        hip.gemm_splitk(a, w, work, ks, out=out2)
    e.record(); torch.cuda.synchronize()
    t2 = s.elapsed_time(e) / reps * 1e-3
    print(f"split-K {ks}: {t2*1e3:.3f} ms {2.0*M*N*K/t2/1e12:.1f} TFLOP/s  max diff vs plain {(out2.float()-out.float()).abs().max().item():.4f}")
print(f"gemm {M}x{N}x{K} tile={os.environ.get('VIS_GEMM_TILE','auto')}: {t*1e3:.3f} ms {2.0*M*N*K/t/1e12:.1f} TFLOP/s")
