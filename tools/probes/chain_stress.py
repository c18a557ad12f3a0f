"""Stress of the chained layer-head launch (csrc/decode_chain.hip): N launches back to back at Qwen2-VL-7B head shapes on
rotating weights and context lengths while a second stream keeps the memory system busy (1 GiB copies) - every launch's y
must equal the four-launch form's bit for bit, the launch counter must advance once per launch and the status word stay 0.
   python tools/probes/chain_stress.py [launches]"""
import os, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vision_inspection_system_amd import hip
dev = torch.device("cuda:0")
hip.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
Hq, Hkv, HD, K, T = 28, 4, 128, 3584, 4096
ns = T // 64
nq = (Hq + 2 * Hkv) * HD
L, C = 4, [0, 63, 64, 700, 2300, 2303, 4000, T - 1]
g = torch.Generator(device=dev).manual_seed(1)
rn = lambda *s, sc=1.0: (torch.randn(s, generator=g, device=dev) * sc).to(torch.bfloat16)
W = [(rn(nq, K, sc=K ** -0.5), rn(nq, sc=0.1), (1 + 0.1 * torch.randn(K, generator=g, device=dev)).to(torch.bfloat16),
      rn(K, Hq * HD, sc=(Hq * HD) ** -0.5)) for _ in range(L)]
kc0, vc0 = rn(Hkv, T, HD), rn(Hkv, T, HD)
ang = torch.rand((T, HD // 2), generator=g, device=dev) * 6.28
emb = torch.cat((ang, ang), -1)
cos_t, sin_t = emb.cos().contiguous(), emb.sin().contiguous()
xs = [rn(K) for _ in range(3)]
steps = [torch.tensor([c], dtype=torch.int32, device=dev) for c in C]
po = torch.empty(Hq * ns * HD, dtype=torch.float32, device=dev)
pml = torch.empty(Hq * ns * 2, dtype=torch.float32, device=dev)
ref = {}
for li in range(L):
    wq, bq, nw, wo = W[li]
    for ci in range(len(C)):
        for xi in range(3):
            k1, v1 = kc0.clone(), vc0.clone()
            qkv = torch.empty(nq, dtype=torch.bfloat16, device=dev)
            att = torch.empty(Hq * HD, dtype=torch.bfloat16, device=dev)
            y = torch.empty(K, dtype=torch.bfloat16, device=dev)
            hip.gemv(xs[xi], wq, qkv, bias=bq, norm_w=nw, eps=1e-6)
            hip.decode_attn(qkv, cos_t, sin_t, k1, v1, steps[ci], po, pml, att, Hq, Hkv, HD, ns, HD ** -0.5)
            hip.gemv(att, wo, y, residual=xs[xi])
            ref[(li, ci, xi)] = y
torch.cuda.synchronize()
ws, sync = hip.decode_chain_state(dev, Hq, Hkv, ns)
# one cache per context length: a launch rewrites only the row of ITS position (which it never reads back from memory), so every
# other row stays what the reference saw
kcs, vcs = [kc0.clone() for _ in C], [vc0.clone() for _ in C]
stop = [False]
a, b = torch.empty(1 << 30, dtype=torch.uint8, device=dev), torch.empty(1 << 30, dtype=torch.uint8, device=dev)


def hog():
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        while not stop[0]:
            for _ in range(8):
                b.copy_(a, non_blocking=True)
            st.synchronize()


th = threading.Thread(target=hog)
th.start()
t0 = time.perf_counter()
B = 512
bad = 0
ys = torch.empty((B, K), dtype=torch.bfloat16, device=dev)
for base in range(0, N, B):
    keys = []
    for i in range(min(B, N - base)):
        n = base + i
        li, ci, xi = n % L, (n // L) % len(C), (n // 7) % 3
        wq, bq, nw, wo = W[li]
        hip.decode_chain(xs[xi], wq, bq, nw, wo, ys[i], cos_t, sin_t, kcs[ci], vcs[ci], steps[ci], ws, sync, Hq, Hkv, HD, ns, HD ** -0.5, 1e-6)
        keys.append((li, ci, xi))
    torch.cuda.synchronize()
    for i, k in enumerate(keys):
        if not torch.equal(ys[i], ref[k]):
            bad += 1
    if (base // B) % 8 == 0:
        print(f"{base + len(keys)} launches, {bad} differing, status {int(sync[hip.CHAIN_STATUS_WORD])}, "
              f"counter {int(sync[0])}, {time.perf_counter() - t0:.1f} s", flush=True)
stop[0] = True
th.join()
torch.cuda.synchronize()
print(f"chained layer head: {N} launches under a bandwidth hog: {bad} differing from the four-launch form, status word "
      f"{int(sync[hip.CHAIN_STATUS_WORD])}, launch counter {int(sync[0])} (expected {N})")
assert bad == 0 and int(sync[hip.CHAIN_STATUS_WORD]) == 0 and int(sync[0]) == N
