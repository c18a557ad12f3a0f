"""Pin the oracle (oracle/qwen2vl_ref.py) against vectors recorded from the real transformers
Qwen2-VL modules (tests/golden/gen_qwen2vl_golden.py).  fp32 vs fp32: tight tolerances."""
import json
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, load_golden, oracle_inputs, ref_config


@pytest.fixture(scope="module")
def tiny():
    from vision_inspection_system_amd.config import Qwen2VLConfig
    from vision_inspection_system_amd.weights import synth_state_dict
    cfg = Qwen2VLConfig.tiny()
    return cfg, ref_config(cfg), synth_state_dict(cfg, seed=0)


def test_smart_resize_table():
    from oracle import qwen2vl_ref as R
    from vision_inspection_system_amd.image_processing import smart_resize
    table = json.load(open(os.path.join(GOLDEN, "smart_resize.json")))
    assert len(table) >= 40
    for row in table:
        assert list(R.smart_resize(row["h"], row["w"])) == row["out"], row
        assert list(smart_resize(row["h"], row["w"])) == row["out"], row
    # the two geometries BASELINE.json quotes
    assert R.smart_resize(448, 448) == (448, 448) and R.smart_resize(1024, 1024) == (980, 980)


@pytest.mark.parametrize("case,frames", [("a", ["frame_a"]), ("b", ["frame_b1", "frame_b2"])])
def test_oracle_matches_transformers(tiny, case, frames):
    from oracle import qwen2vl_ref as R
    cfg, rc, sd = tiny
    g = load_golden()
    fr = [g[n] for n in frames]
    pv, grids = oracle_inputs(fr)
    # preprocessing (rescale, CLIP normalise, merge-order patchify, temporal duplication)
    assert [list(x) for x in grids] == g[f"{case}_grid"].tolist()
    assert abs(float(pv.double().sum()) - float(g[f"{case}_pixel_values_sum"][0])) < 1e-2
    np.testing.assert_allclose(pv.numpy()[[0, 5, -1]], g[f"{case}_pixel_values_rows"], atol=2e-6)
    ids = g[f"ids_{case}"].tolist()
    taps = {}
    toks, logits = R.generate(rc, sd, ids, pv, grids, 16, taps=taps)
    np.testing.assert_array_equal(taps["position_ids"].numpy(), g[f"{case}_position_ids"])
    np.testing.assert_allclose(taps["merger"].numpy(), g[f"{case}_image_embeds"], atol=2e-4, rtol=1e-4)
    np.testing.assert_allclose(logits[0].numpy(), g[f"{case}_first_logits"], atol=2e-4, rtol=1e-4)
    assert toks == g[f"{case}_tokens"].tolist()


def test_oracle_text_only_and_eos(tiny):
    from oracle import qwen2vl_ref as R
    cfg, rc, sd = tiny
    toks, logits = R.generate(rc, sd, [256, 72, 105], None, [], 6)
    assert len(toks) == 6 and len(logits) == 6
    stop = toks[2]
    toks2, _ = R.generate(rc, sd, [256, 72, 105], None, [], 6, eos_ids=(stop,))
    assert toks2 == toks[:3]
