"""Per-kernel microbenchmark at Qwen2-VL-7B shapes (one 1024x1024 image: 4900 patches, S=2249).

Developer tool: prints achieved TFLOP/s or GB/s per kernel class so optimisation work is
aimed at the right kernel.  Random (not zero) operands, HIP-event timing, L2/MALL flushed
between repetitions for the HBM-bound kernels.
"""
import argparse
import json
import math
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vision_inspection_system_amd import hip  # noqa: E402


def timeit(fn, reps=10, warm=3, flush=None):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        if flush is not None:
            flush.add_(1.0)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e-3)
    ts.sort()
    return ts[len(ts) // 2]


def rnd(shape, dev, scale=1.0):
    return (torch.randn(shape, device=dev) * scale).to(torch.bfloat16)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    hip.load()
    res = {}
    flush = torch.zeros(512 * 1024 * 1024 // 4, device=dev)  # 512 MiB > MALL

    gemms = {
        "vit_qkv": (4900, 3840, 1280), "vit_proj": (4900, 1280, 1280), "vit_fc1": (4900, 5120, 1280),
        "vit_fc2": (4900, 1280, 5120), "merger0": (1225, 5120, 5120), "merger2": (1225, 3584, 5120),
        "llm_qkv": (2249, 4608, 3584), "llm_o": (2249, 3584, 3584), "llm_gateup": (2249, 37888, 3584),
        "llm_down": (2249, 3584, 18944), "square4k": (4096, 4096, 4096), "square8k": (8192, 8192, 8192),
    }
    for name, (M, N, K) in gemms.items():
        a, w = rnd((M, K), dev), rnd((N, K), dev, 1 / math.sqrt(K))
        act = hip.ACT_SWIGLU if name == "llm_gateup" else hip.ACT_NONE
        out = torch.empty((M, N // 2 if act else N), dtype=torch.bfloat16, device=dev)
        t = timeit(lambda: hip.gemm(a, w, act=act, out=out))
        res[f"gemm_{name}"] = {"ms": t * 1e3, "TFLOPs": 2.0 * M * N * K / t / 1e12}
        del a, w, out

    # attention
    for name, (S, Hq, Hkv, HD, causal) in {"vit_attn": (4900, 16, 16, 80, False),
                                            "llm_attn": (2249, 28, 4, 128, True)}.items():
        q, k = rnd((Hq, S, HD), dev), rnd((Hkv, S, HD), dev)
        ld = (S + 63) // 64 * 64
        vt = rnd((Hkv, HD, ld), dev)
        o = torch.empty((S, Hq * HD), dtype=torch.bfloat16, device=dev)
        work = hip.make_attn_work([(0, S)], causal, dev)
        t = timeit(lambda: hip.attn_prefill(q, k, vt, o, work, causal, HD ** -0.5))
        flops = 4.0 * S * S * HD * Hq * (0.5 if causal else 1.0)
        res[name] = {"ms": t * 1e3, "TFLOPs": flops / t / 1e12}

    # HBM-bound rows
    x = rnd((2249, 3584), dev)
    w = rnd((3584,), dev)
    y = torch.empty_like(x)
    t = timeit(lambda: hip.rmsnorm(x, w, 1e-6, out=y), flush=flush)
    res["rmsnorm_2249x3584"] = {"ms": t * 1e3, "GBs": 2 * x.numel() * 2 / t / 1e9}
    x = rnd((4900, 1280), dev)
    w, b = rnd((1280,), dev), rnd((1280,), dev)
    y = torch.empty_like(x)
    t = timeit(lambda: hip.layernorm(x, w, b, 1e-6, out=y), flush=flush)
    res["layernorm_4900x1280"] = {"ms": t * 1e3, "GBs": 2 * x.numel() * 2 / t / 1e9}

    S, Hq, Hkv, HD = 2249, 28, 4, 128
    qkv = rnd((S, (Hq + 2 * Hkv) * HD), dev)
    cos = torch.rand((S, HD), device=dev)
    sin = torch.rand((S, HD), device=dev)
    q = torch.empty((Hq, S, HD), dtype=torch.bfloat16, device=dev)
    kc = torch.empty((Hkv, 4096, HD), dtype=torch.bfloat16, device=dev)
    vc = torch.empty_like(kc)
    vt = torch.empty((Hkv, HD, 2304), dtype=torch.bfloat16, device=dev)
    t = timeit(lambda: hip.qkv_rope_split(qkv, cos, sin, q, kc, vc, vt, Hq, Hkv, HD), flush=flush)
    res["rope_split_llm"] = {"ms": t * 1e3, "GBs": (2 * qkv.numel() * 2 + Hkv * HD * S * 2) / t / 1e9}

    # decode GEMVs: algorithmic bytes = N*K*2
    for name, (N, K, act) in {"gemv_qkv": (4608, 3584, 0), "gemv_o": (3584, 3584, 0),
                               "gemv_gateup": (37888, 3584, 3), "gemv_down": (3584, 18944, 0),
                               "gemv_lm_head": (152064, 3584, 0)}.items():
        w = rnd((N, K), dev, 1 / math.sqrt(K))
        xv = rnd((K,), dev)
        nw = rnd((K,), dev)
        out = torch.empty((N // 2 if act else N,), dtype=torch.float32 if name == "gemv_lm_head" else torch.bfloat16,
                          device=dev)
        t = timeit(lambda: hip.gemv(xv, w, out, norm_w=nw if name in ("gemv_qkv", "gemv_gateup") else None, act=act),
                   flush=flush)
        res[name] = {"ms": t * 1e3, "GBs": N * K * 2 / t / 1e9}
        del w

    # fused decode attention at ctx 2300 (rope + KV append + attention + combine)
    kc = rnd((4, 4096, 128), dev)
    vc = rnd((4, 4096, 128), dev)
    qkvd = rnd((36 * 128,), dev)
    cos_t = torch.rand((4096, 128), device=dev)
    sin_t = torch.rand((4096, 128), device=dev)
    step = torch.full((1,), 2300, dtype=torch.int32, device=dev)
    ns = 32
    po = torch.empty(28 * ns * 128, dtype=torch.float32, device=dev)
    pml = torch.empty(28 * ns * 2, dtype=torch.float32, device=dev)
    od = torch.empty(28 * 128, dtype=torch.bfloat16, device=dev)
    t = timeit(lambda: hip.decode_attn(qkvd, cos_t, sin_t, kc, vc, step, po, pml, od, 28, 4, 128, ns, 128 ** -0.5))
    res["decode_attn_ctx2300"] = {"ms": t * 1e3, "GBs": 2 * 4 * 2301 * 128 * 2 / t / 1e9}

    # stream-copy calibration (achievable HBM rate on this box)
    a = torch.empty(1 << 28, dtype=torch.float32, device=dev)
    b = torch.empty_like(a)
    t = timeit(lambda: b.copy_(a))
    res["torch_copy_1GiB"] = {"ms": t * 1e3, "GBs": 2 * a.numel() * 4 / t / 1e9}

    for k_, v in res.items():  # noqa
        print(f"{k_:24s} " + "  ".join(f"{kk}={vv:9.3f}" for kk, vv in v.items()))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
