"""The prompt pass's row kernels at production shapes, cold caches (a 512 MiB flush between repetitions): achieved HBM rate
of RMSNorm / LayerNorm, the split-K finalisation fused with the next norm, and the rope / split kernels (VERDICT r4 item 2b:
"six row passes, 2.64 ms, run at 4 - 4.6 TB/s: get them to >= 5.5").  Prints one line per kernel; --out writes JSON."""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vision_inspection_system_amd import hip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--out", default="")
a = ap.parse_args()
dev = torch.device("cuda:0")
hip.load()
flush = torch.zeros(512 * 1024 * 1024 // 4, device=dev)


def timeit(fn, reps=12, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        flush.add_(1.0)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e-3)
    ts.sort()
    return ts[len(ts) // 2]


def rnd(shape, scale=1.0):
    return (torch.randn(shape, device=dev) * scale).to(torch.bfloat16)


res = {}


def rec(name, t, nbytes):
    res[name] = {"us": t * 1e6, "MB": nbytes / 1e6, "TBps": nbytes / t / 1e12}
    print(f"{name:44s} {t * 1e6:8.1f} us  {nbytes / 1e6:7.1f} MB  {nbytes / t / 1e12:5.2f} TB/s")


for name, (M, N, ln) in {"layernorm 4900x1280 (ViT norm2)": (4900, 1280, True), "rmsnorm 2249x3584 (LLM ln2)": (2249, 3584, False),
                         "rmsnorm 1289x3584 (suffix rows)": (1289, 3584, False)}.items():
    x, w, b = rnd((M, N)), rnd((N,)), rnd((N,))
    y = torch.empty_like(x)
    t = timeit((lambda: hip.layernorm(x, w, b, 1e-6, out=y)) if ln else (lambda: hip.rmsnorm(x, w, 1e-6, out=y)))
    rec(name, t, 2 * M * N * 2)
for name, (M, N, ln) in {"finalize+layernorm 4900x1280 x2 slices (ViT fc2)": (4900, 1280, True),
                         "finalize+rmsnorm 2249x3584 x2 slices (LLM down)": (2249, 3584, False),
                         "finalize+rmsnorm 1289x3584 x2 slices (suffix)": (1289, 3584, False)}.items():
    work = torch.randn((2, M, N), device=dev)
    r, w, b = rnd((M, N)), rnd((N,)), rnd((N,))
    xo, yo = torch.empty_like(r), torch.empty_like(r)
    t = timeit(lambda: hip.splitk_finalize_norm(work, 2, xo, residual=r, norm_w=w, norm_b=b if ln else None, y_out=yo, eps=1e-6))
    rec(name, t, 2 * M * N * 4 + 3 * M * N * 2)
for name, (S, Hq, Hkv, HD, llm) in {"qkv_rope_split<80> 4900 x 16 heads (ViT)": (4900, 16, 16, 80, False),
                                     "qkv_rope_split<128> 2249 x 28/4 heads (LLM)": (2249, 28, 4, 128, True)}.items():
    qkv = rnd((S, (Hq + 2 * Hkv) * HD))
    cos, sin = torch.rand((S, HD), device=dev), torch.rand((S, HD), device=dev)
    q = torch.empty((Hq, S, HD), dtype=torch.bfloat16, device=dev)
    T = 4096 if llm else S
    k = torch.empty((Hkv, T, HD), dtype=torch.bfloat16, device=dev)
    v = torch.empty_like(k) if llm else None
    vt = torch.empty((Hkv, HD, (S + 63) // 64 * 64), dtype=torch.bfloat16, device=dev)
    t = timeit(lambda: hip.qkv_rope_split(qkv, cos, sin, q, k, v, vt, Hq, Hkv, HD))
    nbytes = qkv.numel() * 2 * 2 + Hkv * HD * S * 2 * (1 if llm else 0) + 2 * S * HD * 4
    rec(name, t, nbytes)
if a.out:
    json.dump(res, open(a.out, "w"), indent=1)
