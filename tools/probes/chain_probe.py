"""Where the chained layer-head launch's time goes (tools/probes/chain_probe.sh builds the probe library): every workgroup
stamps the 100 MHz clock at its phase boundaries.  Qwen2-VL-7B head shapes, context 2300 of 4096, 8 different layers'
weights in rotation (cold weights as in the decode step).   python tools/probes/chain_probe.py [ctx]"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libchain_probe.so"))
fn = lib.vis_decode_chain
fn.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_int] + [ctypes.c_void_p] * 12 + [ctypes.c_int] * 8 + [ctypes.c_float] * 2 + [ctypes.c_void_p]
fn.restype = ctypes.c_int
lib.vis_decode_chain_ws_bytes.restype = ctypes.c_longlong
lib.vis_decode_chain_set_probe.argtypes = [ctypes.c_void_p]
Hq, Hkv, HD, K, T = 28, 4, 128, 3584, 4096
ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 2300
ns = T // 64
nq = (Hq + 2 * Hkv) * HD
L = 8
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s, sc=1.0: (torch.randn(s, generator=g, device=dev) * sc).to(torch.bfloat16)
W = [(rn(nq, K, sc=K ** -0.5), rn(nq, sc=0.1), rn(K) * 0 + 1, rn(K, Hq * HD, sc=(Hq * HD) ** -0.5), rn(Hkv, T, HD), rn(Hkv, T, HD))
     for _ in range(L)]
x, y = rn(K), rn(K)
tab = torch.rand((T, HD), generator=g, device=dev)
step = torch.tensor([ctx], dtype=torch.int32, device=dev)
ws = torch.zeros(lib.vis_decode_chain_ws_bytes(Hq, Hkv, ns) // 8, dtype=torch.int64, device=dev)
sync = torch.zeros(64, dtype=torch.int32, device=dev)
n_gv, n_att = nq // 8, Hkv * ns
grid = n_gv + n_att + Hq
probe = torch.zeros((grid, 8), dtype=torch.int64, device=dev)
hog = torch.empty(256 << 20, dtype=torch.uint8, device=dev)

def launch(i):
    wq, bq, nw, wo, kc, vc = W[i % L]
    rc = fn(x.data_ptr(), None, 0, wq.data_ptr(), bq.data_ptr(), nw.data_ptr(), wo.data_ptr(), y.data_ptr(), tab.data_ptr(), tab.data_ptr(),
            kc.data_ptr(), vc.data_ptr(), step.data_ptr(), ws.data_ptr(), sync.data_ptr(), Hq, Hkv, HD, K, K, Hq * HD, T, ns,
            HD ** -0.5, 1e-6, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc

for i in range(16):
    launch(i)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for i in range(64):
    launch(i)
e.record()
torch.cuda.synchronize()
print(f"chained head, ctx {ctx}: {s.elapsed_time(e) / 64 * 1e3:.2f} us per launch (back-to-back eager launches, probe build without stamps armed)")
lib.vis_decode_chain_set_probe(probe.data_ptr())
res = []
for i in range(24):
    hog.fill_(i)                      # cold caches, as between the layers of a decode step
    probe.zero_()
    launch(i)
    torch.cuda.synchronize()
    res.append(probe.cpu().numpy().astype(np.float64) / 100.0)      # us
assert int(sync[32]) == 0
active = min((ctx + 1 + 63) // 64, ns)


def table(name, rows, labels):
    print(f"  {name}: {rows.shape[1]} workgroups")
    for k, lab in labels:
        v = rows[:, :, k]
        v = v[v > 0]
        if v.size:
            v = np.sort(v)
            print(f"    {lab:44s} p10 {v[int(len(v) * .1)]:6.2f}  p50 {v[len(v) // 2]:6.2f}  p90 {v[int(len(v) * .9)]:6.2f}  max {v[-1]:6.2f}")


R = np.stack(res[4:])                                  # [launch, wg, stamp]
t0 = np.where(R[:, :, 0] > 0, R[:, :, 0], np.inf).min(axis=1)  # first workgroup start of each launch
Rz = np.where(R > 0, R - t0[:, None, None], 0.0)
end = Rz.max(axis=(1, 2))
print(f"launch span (first workgroup start -> last stamp): p50 {np.median(end):.2f} us, min {end.min():.2f}, max {end.max():.2f}")
table("projection", Rz[:, :n_gv], [(0, "start"), (1, "x normalised (weights in flight)"), (2, "qkv pair stored"),
                                   (6, "qkv cue seen, W_o requested"), (3, "cue seen (all heads merged)"),
                                   (4, "attention row staged"), (5, "y stored")])
att = Rz[:, n_gv:n_gv + Hkv * active]
table("attention (active splits)", att, [(0, "start"), (1, "K / V requested, position known"), (2, "q / k / v collected"),
                                         (6, "rope, append, scores done"), (7, "softmax done"), (3, "P V done"), (5, "partials stored")])
table("merge", Rz[:, n_gv + n_att:], [(0, "start"), (1, "partials + statistics collected"), (2, "weights exchanged"),
                                      (3, "head stored")])
