"""Row f2: pin oracle/mllama_ref.py against vectors recorded from the real transformers 5.15 mllama modules
(tests/golden/gen_mllama_golden.py -> mllama_tiny.npz).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import mllama_ref as R
from vision_inspection_system_amd.mllama_weights import MllamaConfig, synth_state_dict, tensor_shapes

HERE = os.path.dirname(os.path.abspath(__file__))


def ref_cfg(c: MllamaConfig) -> R.MllamaRefConfig:
    return R.MllamaRefConfig(hidden=c.hidden, layers=c.layers, heads=c.heads, kv_heads=c.kv_heads,
                             intermediate=c.intermediate, vocab=c.vocab, rms_eps=c.rms_eps, rope_theta=c.rope_theta,
                             rope_factor=c.rope_factor, rope_low_freq=c.rope_low_freq, rope_high_freq=c.rope_high_freq,
                             rope_orig_ctx=c.rope_orig_ctx, cross_layers=c.cross_layers, image_token_id=c.image_token_id,
                             v_hidden=c.v_hidden, v_heads=c.v_heads, v_layers=c.v_layers,
                             v_global_layers=c.v_global_layers, v_mlp=c.v_mlp, v_inter=c.v_inter, v_eps=c.v_eps,
                             image_size=c.image_size, patch=c.patch, max_tiles=c.max_tiles)


@pytest.fixture(scope="module")
def tiny():
    cfg = MllamaConfig.tiny()
    return cfg, ref_cfg(cfg), synth_state_dict(cfg, 0), np.load(os.path.join(HERE, "golden", "mllama_tiny.npz"))


def test_canvas_and_fit_match_processor(tiny):
    _, _, _, g = tiny
    for h, w, ch, cw, nh, nw in g["canvas_cases"].tolist():
        assert R.optimal_canvas(h, w, 4, 560) == (ch, cw), (h, w)
        assert R.fit_to_canvas(h, w, ch, cw, 560) == (nh, nw), (h, w)


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_preprocess_matches_processor(tiny, case):
    cfg, rc, _, g = tiny
    tiles, n, (th, tw), ar = R.preprocess_u8(g[f"{case}_image"], rc.image_size, rc.max_tiles)
    assert n == int(g[f"{case}_n_tiles"]) and ar == int(g[f"{case}_ar_id"]) and th * tw == n
    assert np.allclose(tiles.reshape(-1)[::97], g[f"{case}_pixel_sample"], atol=1e-6)
    assert abs(float(tiles.astype(np.float64).sum()) - float(g[f"{case}_pixel_sum"])) < 1e-2


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_oracle_matches_transformers(tiny, case):
    cfg, rc, sd, g = tiny
    taps = {}
    toks, logits = R.generate(rc, sd, g[f"{case}_ids"].tolist(), g[f"{case}_image"], 12, taps=taps)
    cross = taps["cross_states"].numpy()
    assert cross.shape == g[f"{case}_cross_states"].shape
    assert np.abs(cross - g[f"{case}_cross_states"]).max() < 2e-4
    got = torch.stack(logits).numpy()
    assert np.abs(got - g[f"{case}_logits"]).max() < 2e-3
    assert toks == g[f"{case}_tokens"].tolist()


def test_masked_rows_and_shapes(tiny):
    cfg, rc, sd, g = tiny
    assert R.masked_rows(rc, g["a_ids"].tolist()) == 3 and R.masked_rows(rc, g["b_ids"].tolist()) == 12
    assert R.masked_rows(rc, [1, 2, 3]) == 3
    assert set(sd) == set(tensor_shapes(cfg)) and all(tuple(sd[k].shape) == v for k, v in tensor_shapes(cfg).items())
    big = tensor_shapes(MllamaConfig.mllama_11b())
    n = sum(int(np.prod(s)) for s in big.values())
    assert 10.5e9 < n < 10.8e9          # Llama-3.2-11B-Vision: 10.67 B parameters


def test_product_host_logic_matches_processor_and_oracle(tiny):
    """The engine's own host-side geometry / rope tables (it may not import the oracle) against the recorded
    processor decisions and the oracle's tables."""
    from vision_inspection_system_amd import mllama_engine as E
    cfg, rc, _, g = tiny
    for h, w, ch, cw, nh, nw in g["canvas_cases"].tolist():
        assert E.optimal_canvas(h, w, 4, 560) == (ch, cw), (h, w)
        assert E.fit_to_canvas(h, w, ch, cw, 560) == (nh, nw), (h, w)
    assert E.supported_aspect_ratios(4) == R.supported_aspect_ratios(4)
    for c, r in ((cfg, rc), (MllamaConfig.mllama_11b(), ref_cfg(MllamaConfig.mllama_11b()))):
        cos, sin = E.llama3_rope_tables(c, 300)
        rcos, rsin = R.rope_cos_sin(r, torch.arange(300))
        assert np.abs(cos - rcos.numpy()).max() < 1e-4 and np.abs(sin - rsin.numpy()).max() < 1e-4   # f32 pow: numpy vs torch
