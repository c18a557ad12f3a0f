"""Debug probe for vis_decode_proj_fp8's block-scale plumbing: one tile, one K-step, all-ones data, distinct scales per block."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vision_inspection_system_amd import hip
dev = torch.device("cuda:0")
for K in (128, 256):
    N, B = 128, 3
    for live in range(K // 32):
        xq = torch.full((B, K), 0x38, dtype=torch.uint8, device=dev)
        xs = torch.full((B, K // 32), 127, dtype=torch.uint8, device=dev)
        for b in range(B):
            for blk in range(K // 32):
                xs[b, blk] = 127 + blk + 1 + 4 * b          # block blk of row b: 2^(blk + 1 + 4 b)
        wq = torch.zeros((N, K), dtype=torch.uint8, device=dev)
        wq[:, live * 32:(live + 1) * 32] = 0x38               # weights 1.0 in K block `live` only
        sw = torch.ones(N, dtype=torch.float32, device=dev)
        ws = hip.decode_proj_ws(dev, B, N, K, fp8=True)
        out = torch.zeros((B, N), dtype=torch.float32, device=dev)
        hip.decode_proj_fp8(xq, xs, wq, sw, ws, hip.DP_PLAIN, out=out)
        torch.cuda.synchronize()
        exp = [32.0 * 2.0 ** (live + 1 + 4 * b) for b in range(B)]
        print(f"K={K} live block {live}: out[b][0] = {[float(out[b, 0]) for b in range(B)]} expected {exp}; "
              f"uniform over n: {[bool((out[b] == out[b, 0]).all()) for b in range(B)]}")
