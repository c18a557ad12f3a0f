"""PROBE (groundwork for configs[2], DESIGN section 8 "what the next round should take"): do CU-masked streams
(hipExtStreamCreateWithCUMask) give a deterministic split of the chip between an HBM-bound decode projection and an MFMA-bound
prompt-pass GEMM?  Measures, per mask size: the batched decode projection (gate/up shape, 64 rows, stream-K kernel) and the bf16
gate/up GEMM of a 4-image group ALONE on n CUs, then both at the same time on complementary masks from two host threads.
  python tools/probes/cu_mask_probe.py"""
import ctypes, math, os, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vision_inspection_system_amd import hip
dev = torch.device("cuda:0")
hip.load()
rt = ctypes.CDLL("libamdhip64.so")
rt.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
rt.hipExtStreamCreateWithCUMask.restype = ctypes.c_int
NCU = torch.cuda.get_device_properties(0).multi_processor_count


def masked_stream(bits):
    words = (NCU + 31) // 32
    arr = (ctypes.c_uint32 * words)()
    for b in bits:
        arr[b // 32] |= 1 << (b % 32)
    s = ctypes.c_void_p()
    rc = rt.hipExtStreamCreateWithCUMask(ctypes.byref(s), words, arr)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {rc}")
    return torch.cuda.ExternalStream(s.value, device=dev)


B, K, N = 64, 3584, 37888
x = torch.randn((B, K), device=dev).to(torch.bfloat16)
w = (torch.randn((N, K), device=dev) / math.sqrt(K)).to(torch.bfloat16)
part = torch.empty(16 * hip.part_rows(B) * N, dtype=torch.float32, device=dev)
M = 5156
a = torch.randn((M, K), device=dev).to(torch.bfloat16)
out = torch.empty((M, N // 2), dtype=torch.bfloat16, device=dev)


def dec(n):
    for _ in range(n):
        hip.decode_gemm(x, w, part=part)


def gem(n):
    for _ in range(n):
        hip.gemm(a, w, act=hip.ACT_SWIGLU, out=out)


def timed(fn, n, stream):
    with torch.cuda.stream(stream):
        fn(2)
        stream.synchronize()
        t0 = time.perf_counter()
        fn(n)
        stream.synchronize()
        return (time.perf_counter() - t0) / n * 1e6


full = torch.cuda.Stream(device=dev)
d_full, g_full = timed(dec, 40, full), timed(gem, 10, full)
print(f"{NCU} CUs, unmasked stream: decode projection {d_full:.1f} us ({N*K*2/d_full/1e6:.2f} TB/s), gate/up GEMM {g_full:.1f} us", flush=True)
for layout in ("first", "strided"):
    for n in (64, 96, 128, 160, 192, 256):
        bits = list(range(n)) if layout == "first" else sorted(set(int(i * NCU / n) for i in range(n)))
        try:
            st = masked_stream(bits)
        except Exception as e:
            print("mask failed:", e); break
        d, g = timed(dec, 40, st), timed(gem, 10, st)
        print(f"mask {layout:7s} {n:3d} CUs: decode projection {d:7.1f} us = {d_full/d:4.2f} of the full-chip rate, GEMM {g:7.1f} us = {g_full/g:4.2f}", flush=True)
# both at once on complementary masks
for nd in (64, 96, 128):
    sd = masked_stream(list(range(nd)))
    sg = masked_stream(list(range(nd, NCU)))
    res = {}
    def run(name, fn, n, st):
        res[name] = timed(fn, n, st)
    for trial in range(3):
        ts = [threading.Thread(target=run, args=("d", dec, 120, sd)), threading.Thread(target=run, args=("g", gem, 12, sg))]
        t0 = time.perf_counter()
        for t in ts: t.start()
        for t in ts: t.join()
        wall = time.perf_counter() - t0
        print(f"together, decode on {nd} CUs + GEMM on {NCU - nd}: decode {res['d']:7.1f} us = {d_full/res['d']:4.2f} of full rate, GEMM {res['g']:7.1f} us = {g_full/res['g']:4.2f}; "
              f"sum of rates {d_full/res['d'] + g_full/res['g']:4.2f}", flush=True)
