import math, os, sys, torch
sys.path.insert(0, "/root/repo")
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vision_inspection_system_amd import hip
dev = torch.device("cuda:0"); hip.load()
for name, M, N, K in (("llm o x4", 5156, 3584, 3584), ("aud qkv x4", 2816, 6144, 4096), ("llm qkv x4", 5156, 4608, 3584), ("aud o x4?", 2816, 4096, 4096)):
    a = torch.randn((M, K), device=dev).to(torch.bfloat16); w = (torch.randn((N, K), device=dev) / math.sqrt(K)).to(torch.bfloat16)
    r = torch.randn((M, N), device=dev).to(torch.bfloat16); out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    for _ in range(3): hip.gemm(a, w, residual=r, out=out)
    torch.cuda.synchronize(); ts = []
    for _ in range(9):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(4): hip.gemm(a, w, residual=r, out=out)
        e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) / 4 * 1e3)
    ts.sort(); print(f"mix={os.environ.get('VIS_GEMM_MIX1','0'):3s} {name:12s} {M}x{N}x{K}: {ts[len(ts)//2]:7.1f} us", flush=True)
