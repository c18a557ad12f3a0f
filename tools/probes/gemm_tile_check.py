"""Bitwise A/B of a forced GEMM tile kernel against the product's choice: python tools/probes/gemm_tile_check.py dump OUT.pt
(under VIS_GEMM_TILE=<n> or unset), then ... cmp A.pt B.pt.  Shapes: the ragged production ones with their epilogues plus
edge cases (rows / columns that do not fill a tile, K not a multiple of the unroll)."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vision_inspection_system_amd import hip

SHAPES = [("vit proj", 4900, 1280, 1280, "br"), ("llm o", 2249, 3584, 3584, "r"), ("gate/up rest", 2249, 1536, 3584, "s"),
          ("vit qkv", 4900, 3840, 1280, "b"), ("fc1", 4900, 5120, 1280, "bq"), ("ragged", 1291, 1032, 448, "b"),
          ("one k-tile", 300, 520, 64, ""), ("two k-tiles", 257, 512, 128, "r"), ("four k-tiles", 130, 260, 256, "bq"),
          ("five k-tiles", 385, 768, 320, "s"), ("stacked o", 5156, 3584, 3584, "r")]
if sys.argv[1] == "dump":
    dev = torch.device("cuda:0")
    hip.load()
    res = {}
    for name, M, N, K, ep in SHAPES:
        g = torch.Generator(device="cpu").manual_seed(M * 7 + N)
        a = (torch.randn((M, K), generator=g) * 0.5).to(torch.bfloat16).to(dev)
        w = (torch.randn((N, K), generator=g) / math.sqrt(K)).to(torch.bfloat16).to(dev)
        bias = torch.randn((N,), generator=g).to(torch.bfloat16).to(dev) if "b" in ep else None
        act = hip.ACT_SWIGLU if "s" in ep else (hip.ACT_QUICKGELU if "q" in ep else hip.ACT_NONE)
        out = torch.empty((M, N // 2 if act == hip.ACT_SWIGLU else N), dtype=torch.bfloat16, device=dev)
        r = torch.randn(out.shape, generator=g).to(torch.bfloat16).to(dev) if "r" in ep else None
        for _ in range(3):
            hip.gemm(a, w, bias=bias, residual=r, act=act, out=out)
        torch.cuda.synchronize()
        res[name] = out.cpu()
        if act == hip.ACT_NONE:
            ref = a.float() @ w.float().t()
            if bias is not None:
                ref += bias.float()
            if r is not None:
                ref += r.float()
            err = float((out.float() - ref).abs().max())
            print(f"{name:13s} max |err| vs fp32 torch {err:.4f}")
            assert err < 0.06, name
    torch.save(res, sys.argv[2])
else:
    A, B = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    bad = 0
    for k in A:
        same = torch.equal(A[k], B[k])
        print(f"{k:13s} {'bit-identical' if same else 'DIFFERENT: %d elements' % int((A[k] != B[k]).sum())}")
        bad += not same
    sys.exit(1 if bad else 0)
