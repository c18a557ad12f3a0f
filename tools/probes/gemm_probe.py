"""Where a 256x256 ping-pong GEMM tile's time goes (tools/probes/gemm_probe.sh builds the probe library): per workgroup
the 100 MHz clock at entry, main-loop start, main-loop end, stores issued, stores landed, plus the CU it ran on.
Prints, per shape: the launch's span, the distribution of each segment, dispatch skew, and for CUs that ran two
workgroups the hand-over gap.  python tools/probes/gemm_probe.py"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["VIS_GEMM_TILE"] = "7"      # force the ping-pong kernel (read once by the probe library)
dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
here = os.path.dirname(os.path.abspath(__file__))
VARIANTS = {1: "timeline only", 2: "all stores to tile (0,0)", 3: "non-temporal C stores", 4: "no C stores"}

def pct(x):
    x = np.sort(x)
    return f"min {x[0]:6.2f}  p50 {x[len(x)//2]:6.2f}  p90 {x[int(len(x)*.9)]:6.2f}  max {x[-1]:6.2f}"

SHAPES = [("vit fc1 (QuickGELU)", 4900, 5120, 1280, 1), ("vit qkv", 4900, 3840, 1280, 0),
          ("llm gate/up 1278 tiles (SwiGLU)", 2249, 36352, 3584, 3),
          ("llm down split-K 2 (f32 slabs)", 2249, 3584, 18944, -2), ("vit fc2 split-K 2 (f32 slabs)", 4900, 1280, 5120, -2)]
for variant, name, M, N, K, act in [(v,) + s for s in SHAPES for v in sorted(VARIANTS) if v == 1]:
    lib = ctypes.CDLL(os.path.join(here, f"libgemm_probe_{variant}.so"))
    g = lib.vis_gemm_bf16
    g.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 8 + [ctypes.c_void_p]; g.restype = ctypes.c_int
    rd = lib.vis_gemm_probe_read
    rd.argtypes = [ctypes.c_void_p, ctypes.c_int]; rd.restype = ctypes.c_int
    name = f"{name} [{VARIANTS[variant]}]"
    a = torch.randn((M, K), device=dev).to(torch.bfloat16)
    w = (torch.randn((N, K), device=dev) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn((N,), device=dev).to(torch.bfloat16) if act in (1, 2) else None
    out = torch.empty((M, N // 2 if act == 3 else N), dtype=torch.bfloat16, device=dev) if act >= 0 else None
    sk = None
    if act < 0:      # split-K partial launch (f32 slabs): act = -ksplit
        sk = lib.vis_gemm_bf16_splitk_part
        sk.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 6 + [ctypes.c_void_p]; sk.restype = ctypes.c_int
        work = torch.empty(-act * M * N, dtype=torch.float32, device=dev)
    def run():
        if sk is not None:
            rc = sk(a.data_ptr(), w.data_ptr(), work.data_ptr(), M, N, K, a.stride(0), w.stride(0), -act, torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc
            return
        rc = g(a.data_ptr(), w.data_ptr(), bias.data_ptr() if bias is not None else None, None, out.data_ptr(), M, N, K,
               a.stride(0), w.stride(0), out.stride(0), act, 0, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); run(); run(); run(); run(); e.record(); torch.cuda.synchronize()
    ev_us = s.elapsed_time(e) / 4 * 1e3
    nwg = ((M + 255) // 256) * ((N + 255) // 256)      # blockIdx.x range (split-K slices share it: the probe keeps the last writer)
    buf = np.zeros(nwg * 8, dtype=np.uint64)
    assert rd(buf.ctypes.data, nwg * 8) == 0
    t = buf.reshape(nwg, 8)
    ts = (t[:, :5].astype(np.int64) - int(t[:, 0].min())) / 100.0      # us
    hw, xcc = t[:, 6].astype(np.int64), t[:, 7].astype(np.int64) & 15
    cu = ((xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15))
    print(f"== {name}: M={M} N={N} K={K}  {nwg} workgroups  launch-to-launch {ev_us:.1f} us; span entry->last store landed {ts[:, 4].max():.1f} us; "
          f"{len(np.unique(cu))} distinct CUs")
    print(f"   entry time          {pct(ts[:, 0])}")
    print(f"   prologue            {pct(ts[:, 1] - ts[:, 0])}")
    print(f"   main loop           {pct(ts[:, 2] - ts[:, 1])}   ({K // 64} K-steps: {np.median(ts[:, 2] - ts[:, 1]) / (K // 64):.3f} us per step)")
    print(f"   epilogue (issue)    {pct(ts[:, 3] - ts[:, 2])}")
    print(f"   stores landing      {pct(ts[:, 4] - ts[:, 3])}")
    print(f"   end (landed)        {pct(ts[:, 4])}")
    # CUs that ran more than one workgroup: gap between one's end and the next one's entry
    gaps, first_end = [], []
    for c in np.unique(cu):
        idx = np.where(cu == c)[0]
        idx = idx[np.argsort(ts[idx, 0])]
        first_end.append(ts[idx[0], 4])
        for i0, i1 in zip(idx[:-1], idx[1:]):
            gaps.append(ts[i1, 0] - ts[i0, 4])
    if gaps:
        print(f"   hand-over gap (previous workgroup's stores landed -> next entry on the CU) {pct(np.array(gaps))}  [{len(gaps)} hand-overs]")
    # main-loop time by tile row and by XCD (the kernel's blockIdx -> tile map: csrc/common.hip.h xcd_remap)
    tiles_m = (M + 255) // 256
    q, r8 = nwg >> 3, nwg & 7
    bid = np.arange(nwg); x8 = bid & 7
    tid_ = np.where(x8 < r8, x8 * (q + 1), r8 * (q + 1) + (x8 - r8) * q) + (bid >> 3)
    tm = tid_ % tiles_m
    loop = ts[:, 2] - ts[:, 1]
    print("   main loop by tile row : " + "  ".join(f"{i}:{np.median(loop[tm == i]):.1f}" for i in range(tiles_m)))
    print("   main loop by XCD      : " + "  ".join(f"{i}:{np.median(loop[xcc == i]):.1f}" for i in np.unique(xcc)))
    per_cu = np.bincount(np.unique(cu, return_inverse=True)[1])
    print(f"   workgroups per CU: " + ", ".join(f"{k}: {int((per_cu == k).sum())} CUs" for k in np.unique(per_cu)))
