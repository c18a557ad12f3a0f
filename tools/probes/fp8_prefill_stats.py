import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from helpers import load_golden, oracle_inputs, ref_config
from test_engine_gpu import _dequantised_sd
from oracle import qwen2vl_ref as R
from vision_inspection_system_amd.config import Qwen2VLConfig
from vision_inspection_system_amd.engine import Qwen2VLEngine
from vision_inspection_system_amd.weights import pack_device_weights, synth_state_dict
dev = "cuda:0"
cfg = Qwen2VLConfig.tiny(); sd = synth_state_dict(cfg, 0)
ids = [256, 10, 20, 30] + list(range(40, 90)) + [257]; fr = []
pv, grids = None, []
dsd = _dequantised_sd(cfg, sd); psd = dict(dsd); psd["lm_head.weight"] = sd["lm_head.weight"]
t8, t16 = {}, {}
_, l8 = R.generate(ref_config(cfg), sd, ids, pv, grids, 1, prefill_fp8_sd=psd, taps=t8)
_, l16 = R.generate(ref_config(cfg), sd, ids, pv, grids, 1, taps=t16)
for mode in ("fp8", "bf16"):
    eng = Qwen2VLEngine(cfg, pack_device_weights(cfg, sd, dev), dev, max_ctx=256, decode_splits=4, prefill_dtype=mode)
    t = {}
    eng.prefill(ids, [torch.from_numpy(f).to(dev) for f in fr], taps=t)
    got = t["first_logits"].float().cpu()
    for name, ref in (("oracle-fp8", l8[0]), ("oracle-bf16", l16[0])):
        d = (got - ref).abs()
        print(f"engine {mode:4s} vs {name:11s}: mean {d.mean():.4f} max {d.max():.4f}")
    for name, ref in (("oracle-fp8", t8["layer0"]), ("oracle-bf16", t16["layer0"])):
        d = (t["layer0"].float().cpu() - ref).abs()
        print(f"   layer0 {mode:4s} vs {name:11s}: mean {d.mean():.5f} max {d.max():.4f}")
print("oracle fp8 vs oracle bf16: mean %.4f max %.4f" % ((l8[0]-l16[0]).abs().mean(), (l8[0]-l16[0]).abs().max()))
