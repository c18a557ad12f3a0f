"""The logit tolerance AT DEPTH (north_star: "within a stated float tolerance on the logits" of the call
at /root/reference/src/agents/vlm_inspector.py:105-111).

tests/test_fullsize_oracle_gpu.py compares with the fp32 oracle at the exact 7B shapes ONE layer deep; here the same
comparison runs 4 and 8 layers deep (ViT blocks AND decoder layers, variance-preserving weights), for the bf16 engine
against the fp32 oracle and for the fp8 configuration (fp8 MFMA prompt pass, e4m3 decode weights) against the oracle that
fake-quantises the same tensors - and at FULL depth (32 ViT blocks, 28 decoder layers), where the fp32 oracle does not
finish in test time, the fp8 engine is compared with the bf16 engine on the same weights.  Every figure is printed;
DESIGN.md section 2 carries the table.

Stated tolerances, relative to the range R = max |reference logit| (measured r04 in brackets; DESIGN section 2 has the table):
  bf16 vs fp32 oracle   depth 4: max <= 1.5 % [0.67], mean <= 0.25 % [0.11]     depth 8: max <= 2.0 % [0.86], mean <= 0.35 % [0.15]
                        (depth 1, test_fullsize_oracle_gpu.py: max <= 1.5 % [0.33-0.68], mean <= 0.2 % [0.06-0.10])
  fp8  vs fp8 oracle    depth 4: max <= 10 % [4.5],   mean <= 1.6 % [0.83]      depth 8: max <= 12 % [5.6],   mean <= 2.0 % [1.0]
                        (a quantiser is discontinuous: one e4m3 step is 6 %, so the agreement is statistical)
  fp8 vs bf16 engine at FULL depth, rms over the 152 064 logits: <= 6 % of the logits' standard deviation (= 1.3 % of their
  range) on weights whose residual writers are damped by 1 / sqrt(2 L) [3.5-4.0 %]; <= 25 % (5.5 % of range) on undamped
  variance-preserving weights, every branch at unit gain [16-17 %]."""
import dataclasses

import numpy as np
import pytest
import torch

from helpers import dequantised_sd, oracle_inputs, ref_config

pytestmark = pytest.mark.gpu

BF16_TOL = {4: (0.015, 0.0025), 8: (0.020, 0.0035)}
FP8_TOL = {4: (0.10, 0.016), 8: (0.12, 0.020)}
FULL_RMS_OF_STD = {"damped": 0.06, "undamped": 0.25}


def _stat(tag, name, got, ref, rel_max, rel_mean):
    got, ref = got.float().cpu(), ref.float()
    assert got.shape == ref.shape and torch.isfinite(got).all(), name
    R = float(ref.abs().max())
    d = (got - ref).abs()
    mx, mean = float(d.max()) / R, float(d.mean()) / R
    print(f"[{tag}] {name}: range {R:.3f}  max {mx * 100:.3f} %  mean {mean * 100:.4f} % of range")
    assert mx <= rel_max, f"{name}: max error {mx * 100:.2f} % of range (> {rel_max * 100} %)"
    assert mean <= rel_mean, f"{name}: mean error {mean * 100:.3f} % of range (> {rel_mean * 100} %)"
    return float(d.max())


def _pick_ok(name, got, ref, err):
    top2 = torch.topk(ref.float(), 2).values
    if float(top2[0] - top2[1]) > 2 * err:
        assert int(got.float().argmax()) == int(ref.argmax()), f"{name}: greedy pick differs off a near-tie"


def _prompt(cfg, n_image_tokens, n_text=1024, seed=99):
    rng = np.random.default_rng(seed)
    text = rng.integers(0, min(151643, cfg.vocab - 16), n_text - 2).tolist()
    return text[:16] + [cfg.vision_start_id] + [cfg.image_token_id] * n_image_tokens + [cfg.vision_end_id] + text[16:]


@pytest.mark.parametrize("depth", [4, 8])
def test_7b_shapes_depth_vs_oracle(device, depth):
    """bf16 engine vs fp32 oracle and fp8 engine vs fake-quant oracle, `depth` ViT blocks + `depth` decoder layers at the
    exact Qwen2-VL-7B shapes (N = 4900 patches, S = 2249, 152064-row lm_head): image features, first-step logits and three
    teacher-forced decode steps."""
    from oracle import qwen2vl_ref as R
    from vision_inspection_system_amd.config import Qwen2VLConfig
    from vision_inspection_system_amd.engine import Qwen2VLEngine
    from vision_inspection_system_amd.weights import pack_device_weights, synth_state_dict
    cfg = dataclasses.replace(Qwen2VLConfig.qwen2_vl_7b(), layers=depth, v_depth=depth)
    sd = synth_state_dict(cfg, seed=5, rng="torch", device=device)
    w = pack_device_weights(cfg, sd, device)
    frame = np.random.default_rng(21).integers(0, 256, (980, 980, 3), dtype=np.uint8)
    ids = _prompt(cfg, (980 // 14) ** 2 // 4)
    pv, grids = oracle_inputs([frame])
    fr = [torch.from_numpy(frame).to(device)]

    # ---- (a) bf16 engine vs the fp32 oracle
    mx, mean = BF16_TOL[depth]
    eng = Qwen2VLEngine(cfg, w, device, max_ctx=2560)
    taps, rtaps = {}, {}
    eng.prefill(ids, fr, taps=taps, max_new_tokens=8)
    with torch.no_grad():
        ref_toks, ref_logits = R.generate(ref_config(cfg), sd, ids, pv, grids, 4, taps=rtaps)
    tag = f"depth {depth} bf16"
    _stat(tag, "image features [1225, 3584]", taps["image_embeds"], rtaps["merger"], mx, mean)
    err = _stat(tag, "first-step logits", taps["first_logits"], ref_logits[0], mx, mean)
    _pick_ok("first token", taps["first_logits"], ref_logits[0], err)
    for t in range(3):
        eng.cur_token.fill_(ref_toks[t])
        eng.decode(1, use_graph=False)
        err = _stat(tag, f"decode step {t + 1} logits", eng.logits, ref_logits[t + 1], mx, mean)
        _pick_ok(f"decode step {t + 1}", eng.logits, ref_logits[t + 1], err)
    logits16 = ref_logits
    del eng
    torch.cuda.empty_cache()

    # ---- (b) fp8 engine vs the oracle that fake-quantises the same tensors (weights: the engine's e4m3 codes de-quantised;
    # activations: per-row e4m3 in front of every projection; the first token's lm_head and the merger stay bf16)
    mx, mean = FP8_TOL[depth]
    eng = Qwen2VLEngine(cfg, w, device, max_ctx=2560, prefill_dtype="fp8", decode_weights="fp8")
    assert eng.vq8, "the fp8 configuration runs the ViT block projections in e4m3 as well"
    dsd = {k: v.cpu() for k, v in dequantised_sd(cfg, sd).items()}
    psd = dict(dsd)
    psd["lm_head.weight"] = sd["lm_head.weight"]
    taps, r8 = {}, {}
    eng.prefill(ids, fr, taps=taps, max_new_tokens=8)
    with torch.no_grad():
        toks8, logits8 = R.generate(ref_config(cfg), sd, ids, pv, grids, 4, taps=r8, prefill_fp8_sd=psd, decode_sd=dsd)
    tag = f"depth {depth} fp8"
    _stat(tag, "image features (fp8 tower)", taps["image_embeds"], r8["merger"], mx, mean)
    _stat(tag, "first-step logits", taps["first_logits"], logits8[0], mx, mean)
    d8 = float((taps["first_logits"].float().cpu() - logits8[0]).abs().mean())
    d16 = float((taps["first_logits"].float().cpu() - logits16[0]).abs().mean())
    print(f"[{tag}] first logits: mean |d| to the fp8 oracle {d8:.4f}, to the fp32 oracle {d16:.4f}")
    assert d8 < d16, "the fp8 engine must be closer to the fake-quant oracle than to the unquantised one"
    for t in range(3):
        eng.cur_token.fill_(toks8[t])
        eng.decode(1, use_graph=False)
        _stat(tag, f"decode step {t + 1} logits (e4m3 weights)", eng.logits, logits8[t + 1], mx, mean)
    del eng
    torch.cuda.empty_cache()


@pytest.mark.parametrize("family", ["damped", "undamped"])
def test_7b_full_depth_fp8_vs_bf16(device, family):
    """FULL depth (32 ViT blocks + 28 decoder layers, exact 7B shapes): the fp8 configuration against the bf16 engine on the
    same variance-preserving weights - first-step logits and three teacher-forced decode steps.  "damped": the matrices that
    write into the residual stream carry 1 / sqrt(2 L) (how trained transformers are initialised); "undamped": every branch
    at unit gain - the most perturbation-sensitive network that still keeps O(1) activations.  What the bound means: e4m3
    has three mantissa bits, i.e. ~3.6 % of relative rms noise per GEMM output (both operands quantised); the full-depth
    error is that figure carried through 60 layers.  On the FLAT N(0, 0.02) benchmark weights (per-layer gain 1.4 - 7.6) ANY
    perturbation grows with depth - tools/depth_error.py: fp8 vs bf16 rms 11 % of the logit std at depth 1, 51 % at depth 28,
    and bf16's own rounding is amplified by the same factor - which is what the r03 run of this comparison (22 % of range)
    had measured: a property of those weights, not of the quantisation scheme; the r03 docstring's "uncorrelated" was
    wrong (correlation 0.87) and is withdrawn."""
    from vision_inspection_system_amd.config import Qwen2VLConfig
    from vision_inspection_system_amd.engine import Qwen2VLEngine
    from vision_inspection_system_amd.weights import random_device_weights
    cfg = Qwen2VLConfig.qwen2_vl_7b()
    w = random_device_weights(cfg, device, seed=3, scaled=True,
                              branch_gain=(2 * cfg.layers) ** -0.5 if family == "damped" else 1.0)
    frame = torch.from_numpy(np.random.default_rng(21).integers(0, 256, (980, 980, 3), dtype=np.uint8)).to(device)
    ids = _prompt(cfg, (980 // 14) ** 2 // 4)
    e16 = Qwen2VLEngine(cfg, w, device, max_ctx=2560)
    e8 = Qwen2VLEngine(cfg, w, device, max_ctx=2560, prefill_dtype="fp8", decode_weights="fp8")
    t16, t8 = {}, {}
    e16.prefill(ids, [frame], taps=t16, max_new_tokens=8)
    e8.prefill(ids, [frame], taps=t8, max_new_tokens=8)
    pairs = [("first-step logits", t8["first_logits"].float(), t16["first_logits"].float())]
    for t in range(3):
        tok = int(e16.logits.float().argmax())
        for e in (e16, e8):
            e.cur_token.fill_(tok)
            e.decode(1, use_graph=False)
        pairs.append((f"decode step {t + 1}", e8.logits.float().clone(), e16.logits.float().clone()))
    bound = FULL_RMS_OF_STD[family]
    for name, a, b in pairs:
        assert torch.isfinite(a).all()
        sd_, rng_ = float(b.std()), float(b.abs().max())
        rms, mx = float((a - b).pow(2).mean().sqrt()), float((a - b).abs().max())
        top5 = len(set(torch.topk(a, 5).indices.tolist()) & set(torch.topk(b, 5).indices.tolist()))
        print(f"[full depth fp8 vs bf16, {family}] {name}: logit std {sd_:.3f} range {rng_:.3f}  rms {rms / sd_ * 100:.2f} % of std = "
              f"{rms / rng_ * 100:.2f} % of range, max {mx / rng_ * 100:.2f} % of range, "
              f"top-1 {'same' if int(a.argmax()) == int(b.argmax()) else 'differs'}, top-5 overlap {top5}")
        assert rms <= bound * sd_, f"{name}: rms {rms / sd_ * 100:.1f} % of the logit std (> {bound * 100} %)"
        assert mx <= 6.0 * rms, f"{name}: outlier {mx / rms:.1f} x rms"      # noise-like: no systematic channel error
        assert top5 >= (3 if family == "damped" else 1), f"{name}: top-5 sets overlap in {top5}"
        top2 = torch.topk(b, 2).values
        if float(top2[0] - top2[1]) > 2 * mx:
            assert int(a.argmax()) == int(b.argmax()), f"{name}: greedy pick differs off a near-tie"
    del e16, e8
    torch.cuda.empty_cache()


@pytest.mark.parametrize("depth", [4, 8])
def test_mllama_11b_shapes_depth_vs_oracle(device, depth):
    """The Auditor's logit tolerance at depth (the call /root/reference/src/agents/vlm_auditor.py:152-158
    becomes): Llama-3.2-11B-Vision shapes, `depth` vision layers (the last quarter global / gated, intermediate features
    concatenated as in the released model) over the 2 x 2-tile canvas of a 1024^2 image, projector, `depth` decoder layers of
    which index 3 is a cross-attention layer (the released model's first: cross_attention_layers = [3, 8, ...]), the 128 256-row
    lm_head; bf16 engine against the fp32 oracle (oracle/mllama_ref.py, pinned to transformers by tests/test_oracle_mllama.py):
    cross states, the last layer's hidden state, first-step logits and three teacher-forced decode steps on the KV cache and the
    cached cross keys.  Same stated bounds as the Inspector's rows of the table in DESIGN section 2."""
    from oracle import mllama_ref as R
    from test_oracle_mllama import ref_cfg
    from vision_inspection_system_amd.mllama_engine import MllamaEngine
    from vision_inspection_system_amd.mllama_weights import MllamaConfig, pack_device_weights, synth_state_dict
    n_glob = depth // 4
    n_loc = depth - n_glob
    cfg = dataclasses.replace(MllamaConfig.mllama_11b(), layers=depth, cross_layers=(3,), v_layers=n_loc,
                              v_global_layers=n_glob, v_inter=tuple(range(1, n_loc, 2)))
    sd = synth_state_dict(cfg, seed=6, rng="torch", device=device)
    eng = MllamaEngine(cfg, pack_device_weights(cfg, sd, device), device, max_ctx=1024)
    rng = np.random.default_rng(3)
    image = rng.integers(0, 256, (1024, 1024, 3), dtype=np.uint8)
    ids = [1] + rng.integers(1000, cfg.vocab - 8, 600).tolist() + [cfg.image_token_id] + \
        rng.integers(1000, cfg.vocab - 8, 102).tolist()                      # 704 tokens, image token after the text
    taps, rtaps = {}, {}
    eng.prefill(ids, torch.from_numpy(image).to(device), taps=taps)
    with torch.no_grad():
        ref_toks, ref_logits = R.generate(ref_cfg(cfg), sd, ids, image, 4, taps=rtaps)
    mx, mean = BF16_TOL[depth]
    tag = f"mllama depth {depth} bf16"
    _stat(tag, f"vision tower ({n_loc} local + {n_glob} global layers) + projector -> cross states", taps["cross_states"],
          rtaps["cross_states"], mx, mean)
    _stat(tag, f"hidden state after decoder layer {depth - 1} (layer 3 = cross-attention), S = 704", taps[f"layer{depth - 1}"],
          rtaps[f"layer{depth - 1}"], mx, mean)
    err = _stat(tag, "first-step logits [128256]", taps["first_logits"], ref_logits[0], mx, mean)
    _pick_ok("first token", taps["first_logits"], ref_logits[0], err)
    for t in range(3):
        eng.cur_token.fill_(ref_toks[t])
        eng.decode(1, use_graph=False)
        err = _stat(tag, f"decode step {t + 1} logits", eng.logits, ref_logits[t + 1], mx, mean)
        _pick_ok(f"decode step {t + 1}", eng.logits, ref_logits[t + 1], err)
    del eng
    torch.cuda.empty_cache()
