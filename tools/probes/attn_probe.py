"""Times the prefill attention kernel built with each ATT_PROBE bit set (tools/probes/attn_probe.sh builds the
libraries): which of K/V traffic, the transcendental, the two MFMA groups and the per-tile barrier the kernel's time is
made of.  Probe results are wrong by construction; only durations are printed."""
import ctypes, glob, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vision_inspection_system_amd import hip
dev = torch.device("cuda:0")
hip.load()
here = os.path.dirname(os.path.abspath(__file__))
NAMES = {0: "baseline", 1: "no K/V traffic after tile 1", 2: "exp -> mul", 4: "no PV MFMA", 8: "no QK MFMA", 16: "no barrier",
         3: "no traffic + exp->mul", 6: "exp->mul + no PV", 7: "no traffic, exp->mul, no PV"}
for which in ("vit", "llm"):
    S, Hq, Hkv, HD, causal = (4900, 16, 16, 80, False) if which == "vit" else (2249, 28, 4, 128, True)
    q = torch.randn((Hq, S, HD), device=dev).to(torch.bfloat16)
    k = torch.randn((Hkv, S, HD), device=dev).to(torch.bfloat16)
    ld = (S + 63) // 64 * 64
    vt = torch.randn((Hkv, HD, ld), device=dev).to(torch.bfloat16)
    o = torch.empty((S, Hq * HD), dtype=torch.bfloat16, device=dev)
    work = hip.make_attn_work([(0, S)], causal, dev, heads=Hq if not causal else 0)
    for path in sorted(glob.glob(os.path.join(here, "libattn_probe_*.so")), key=lambda p: int(p.split("_")[-1][:-3])):
        bits = int(path.split("_")[-1][:-3])
        lib = ctypes.CDLL(path)
        f = lib.vis_attn_prefill_rows
        f.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 9 + [ctypes.c_float, ctypes.c_int, ctypes.c_void_p]
        f.restype = ctypes.c_int
        args = (q.data_ptr(), k.data_ptr(), vt.data_ptr(), o.data_ptr(), work.data_ptr(), work.shape[0], Hq, Hkv, HD, S, S,
                ld, o.stride(0), 1 if causal else 0, HD ** -0.5, 0, torch.cuda.current_stream().cuda_stream)
        for _ in range(3):
            assert f(*args) == 0
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            f(*args)
        e.record(); torch.cuda.synchronize()
        print(f"{which} probe {bits:2d} ({NAMES.get(bits, '?'):32s}): {s.elapsed_time(e) / 20 * 1e3:8.1f} us")
