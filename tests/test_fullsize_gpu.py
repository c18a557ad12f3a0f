"""End-to-end properties at BASELINE's FULL size (exact Qwen2-VL-7B shapes, 980x980 frames, S = 1225 image tokens +
text), where the fp32 oracle cannot run in test time.  The oracle parity proper is at the tiny configuration
(test_engine_gpu.py); here the size-independent properties of the path are checked on seeded random bf16 weights:
reproducibility, batch invariance (a request's answer does not depend on what shares the batch with it), hipGraph
replay == eager launches, and agreement between the two arithmetic paths that compute the same function - the prompt
pass (MFMA GEMMs + flash attention) and the decode step (GEMVs + split-context cache attention)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big(device):
    from vision_inspection_system_amd.config import Qwen2VLConfig
    from vision_inspection_system_amd.engine import Qwen2VLEngine
    from vision_inspection_system_amd.weights import random_device_weights
    cfg = Qwen2VLConfig.qwen2_vl_7b()
    eng = Qwen2VLEngine(cfg, random_device_weights(cfg, device, 0), device, max_ctx=4096, max_batch=4)
    rng = np.random.default_rng(7)
    frames = [torch.from_numpy(rng.integers(0, 256, (980, 980, 3), dtype=np.uint8)).to(device) for _ in range(2)]
    n_img = (980 // 14) ** 2 // 4

    def ids(seed, n_text=200):
        r = np.random.default_rng(seed)
        return [cfg.vision_start_id] + [cfg.image_token_id] * n_img + [cfg.vision_end_id] + r.integers(0, 1000, n_text).tolist()

    reqs = [(ids(0), [frames[0]]), (ids(1, 333), [frames[1]])]
    yield cfg, eng, reqs
    del eng
    torch.cuda.empty_cache()


def test_7b_batch_invariance_and_reproducibility(big):
    cfg, eng, (ra, rb) = big
    single_a = eng.generate(ra[0], ra[1], max_new_tokens=12, ignore_eos=True)
    single_b = eng.generate(rb[0], rb[1], max_new_tokens=12, ignore_eos=True)
    assert len(single_a) == 12 and all(0 <= t < cfg.vocab for t in single_a + single_b)
    assert eng.generate(ra[0], ra[1], max_new_tokens=12, ignore_eos=True) == single_a       # reproducible
    out = eng.generate_batch([ra, rb, ra, rb], max_new_tokens=12, ignore_eos=True)
    # the same request gives the same tokens in any slot of a batch ...
    assert out[0] == out[2], "request A differs between slots 0 and 2"
    assert out[1] == out[3], "request B differs between slots 1 and 3"
    # ... and whatever shares the batch with it (other lengths, other batch size, other slot order)
    other = eng.generate_batch([rb, ra, ra], max_new_tokens=12, ignore_eos=True)
    assert other == [out[1], out[0], out[0]]
    assert eng.generate_batch([ra, rb], max_new_tokens=12, ignore_eos=True) == [out[0], out[1]]
    # against the single-sequence path the prompt pass is the same code (first token exact); the later tokens come
    # from different decode kernels (GEMV vs the batched stream-K projection: another f32 summation order), so they
    # agree only up to near-ties - with random weights the logits are nearly flat and ties are common
    assert out[0][0] == single_a[0] and out[1][0] == single_b[0]


def test_7b_graph_replay_equals_eager(big):
    cfg, eng, (ra, rb) = big
    eager = eng.generate(rb[0], rb[1], max_new_tokens=6, ignore_eos=True, use_graph=False)
    graph = eng.generate(rb[0], rb[1], max_new_tokens=6, ignore_eos=True, use_graph=True)
    assert eager == graph


def test_7b_chained_decode_equals_four_launches(big, monkeypatch):
    """The single-sequence decode step with the head of every layer as ONE chained launch (csrc/decode_chain.hip; the first
    layer reading the token's embedding row itself, the pick's first stage in the lm_head epilogue) against the same step as
    separate launches (VIS_DECODE_CHAIN=0), full depth, exact 7B shapes: every generated token and the last step's 152 064
    logits bit for bit, eager and as a replayed hipGraph, greedy and sampled; the chain's status word stays clean."""
    from vision_inspection_system_amd import hip
    from vision_inspection_system_amd.engine import Qwen2VLEngine
    cfg, eng, (ra, rb) = big
    assert eng.chain_sync is not None, "the chained layer head must be the default decode path at 7B shapes"
    monkeypatch.setenv("VIS_DECODE_CHAIN", "0")
    plain = Qwen2VLEngine(cfg, eng.w, eng.device, max_ctx=4096)
    monkeypatch.delenv("VIS_DECODE_CHAIN")
    assert plain.chain_sync is None
    try:
        for temperature, use_graph in ((0.0, True), (0.0, False), (0.7, True)):
            a = eng.generate(rb[0], rb[1], max_new_tokens=24, ignore_eos=True, use_graph=use_graph, temperature=temperature, seed=11)
            la = eng.logits.clone()
            b = plain.generate(rb[0], rb[1], max_new_tokens=24, ignore_eos=True, use_graph=use_graph, temperature=temperature,
                               seed=11)
            assert a == b, f"tokens differ (temperature {temperature}, graph {use_graph})"
            assert torch.equal(la, plain.logits), "last-step logits differ"
            assert int(eng.chain_sync[hip.CHAIN_STATUS_WORD]) == 0
        assert int(eng.chain_sync[0]) > 0        # launches were counted: the chained path really ran
    finally:
        del plain
        torch.cuda.empty_cache()


def test_7b_two_engines_decoding_on_two_streams(big):
    """Two engines on one GPU, driven by two host threads on two HIP streams at the same time (a host app with two different
    Qwen models, or two agents on their own streams).  A chained launch needs its whole grid resident, so two of them must not
    run side by side: Qwen2VLEngine.decode orders the chained decode calls of a device on the GPU with an event.  Both engines
    must produce exactly their solo tokens, every round, with clean status words."""
    import threading
    from vision_inspection_system_amd import hip
    from vision_inspection_system_amd.engine import Qwen2VLEngine
    cfg, eng, (ra, rb) = big
    other = Qwen2VLEngine(cfg, eng.w, eng.device, max_ctx=4096)
    try:
        assert eng.chain_sync is not None and other.chain_sync is not None
        solo = (eng.generate(ra[0], ra[1], max_new_tokens=40, ignore_eos=True),
                other.generate(rb[0], rb[1], max_new_tokens=40, ignore_eos=True))
        for rnd in range(3):
            out, errs = [None, None], []

            def run(i, e, r):
                try:
                    st = torch.cuda.Stream(device=e.device)
                    with torch.cuda.stream(st):
                        out[i] = e.generate(r[0], r[1], max_new_tokens=40, ignore_eos=True, check_every=4)
                    st.synchronize()
                except Exception as ex:      # noqa: BLE001
                    errs.append(ex)
            ts = [threading.Thread(target=run, args=(0, eng, ra)), threading.Thread(target=run, args=(1, other, rb))]
            for t in ts:
                t.start()
            for t in ts:
                t.join()
            assert not errs, errs
            assert out[0] == solo[0] and out[1] == solo[1], f"round {rnd}: tokens differ from the solo runs"
            torch.cuda.synchronize()
            assert int(eng.chain_sync[hip.CHAIN_STATUS_WORD]) == 0 and int(other.chain_sync[hip.CHAIN_STATUS_WORD]) == 0
    finally:
        del other
        torch.cuda.empty_cache()


def test_7b_decode_step_agrees_with_prompt_pass(big):
    """Logits of the token after (prompt + 3 generated tokens): once from three decode steps on the KV cache, once
    from a prompt pass over the extended prompt.  Same function, two kernel families (GEMV / cache attention vs MFMA
    GEMM / flash attention); bf16 activations, f32 accumulation in both: the logits agree to bf16 rounding noise."""
    cfg, eng, (ra, rb) = big
    toks = eng.generate(ra[0], ra[1], max_new_tokens=4, ignore_eos=True)
    via_decode = eng.logits.float().clone()
    eng.prefill(list(ra[0]) + toks[:3], ra[1], max_new_tokens=2)
    via_prefill = eng.logits.float().clone()
    scale = float(via_prefill.abs().max())
    diff = float((via_decode - via_prefill).abs().max())
    rms = float((via_decode - via_prefill).pow(2).mean().sqrt())
    print(f"decode vs prompt pass: rms {rms / scale:.5f}, max {diff / scale:.5f} of the logit range {scale:.3f}")
    # The criterion is the RMS difference over the 152 064 logits (<= 1 % of the logit range): 28 layers of bf16 hidden
    # states on N(0, 0.02) weights give nearly flat logits (range ~ +-5.5) and an error that is noise-like per logit, so
    # the MAXIMUM is an extreme-value statistic (~4.5 sigma of 152 k samples) that moves by a tenth of itself between
    # builds and boxes with nothing wrong (r01 / r02: 2.6 - 3.0 %, once 3.006 % against a 3 % limit - ADVICE r2); it is kept
    # only as a loose outlier guard (6 sigma).  The exact-shape ORACLE comparison is tests/test_fullsize_oracle_gpu.py.
    assert rms <= 0.010 * scale, f"decode vs prefill logits: rms difference {rms} (scale {scale})"
    assert diff <= 6.0 * max(rms, 1e-6) and diff <= 0.05 * scale, f"decode vs prefill logits: outlier {diff} (rms {rms}, scale {scale})"
    top2 = torch.topk(via_prefill, 2).values
    if float(top2[0] - top2[1]) > 2 * diff:       # not a near-tie: the greedy pick must be the same
        assert int(via_decode.argmax()) == int(via_prefill.argmax()) == toks[3]


def test_7b_shared_text_prefix_is_bit_identical(big, monkeypatch):
    """The reference's message order (text part, then the image): the ~1000-token inspection prompt is common to the
    images of a batch; computing its K / V once per batch must not change a single token or logit."""
    cfg, eng, (ra, rb) = big
    rng = np.random.default_rng(11)
    text = rng.integers(0, 1000, 1000).tolist()
    n_img = (980 // 14) ** 2 // 4
    mk = lambda tail: text + [cfg.vision_start_id] + [cfg.image_token_id] * n_img + [cfg.vision_end_id] + tail
    reqs = [(mk([5, 6, 7]), ra[1]), (mk([8]), rb[1]), (mk([5, 6, 7]), ra[1])]
    assert eng.shared_prefix_len([r[0] for r in reqs]) == 960
    shared = eng.generate_batch(reqs, max_new_tokens=8, ignore_eos=True)
    logits = eng.logits_b[:3].clone()
    monkeypatch.setenv("VIS_SHARE_PREFIX", "0")
    plain = eng.generate_batch(reqs, max_new_tokens=8, ignore_eos=True)
    assert shared == plain and torch.equal(logits, eng.logits_b[:3])
    assert shared[0] == shared[2]


def test_mllama_11b_properties(device):
    """Row f2 at exact Llama-3.2-11B-Vision shapes (seeded random weights, 1024x1024 image = 2x2 tiles): reproducible,
    graph == eager, and the decode step (GEMVs, cache self-attention, cached cross-attention keys) agrees with a prompt
    pass over the extended prompt."""
    from vision_inspection_system_amd.mllama_engine import MllamaEngine
    from vision_inspection_system_amd.mllama_weights import MllamaConfig, random_device_weights
    cfg = MllamaConfig.mllama_11b()
    eng = MllamaEngine(cfg, random_device_weights(cfg, device, 0), device, max_ctx=1024, max_batch=3)
    rng = np.random.default_rng(3)
    frame = torch.from_numpy(rng.integers(0, 256, (1024, 1024, 3), dtype=np.uint8)).to(device)
    ids = [1] + rng.integers(1000, cfg.vocab - 8, 150).tolist() + [cfg.image_token_id] + rng.integers(1000, cfg.vocab - 8, 20).tolist()
    a = eng.generate(ids, frame, max_new_tokens=6, stop_on_eos=False, use_graph=True)
    via_decode = eng.logits.float().clone()
    b = eng.generate(ids, frame, max_new_tokens=6, stop_on_eos=False, use_graph=False)
    assert a == b and len(a) == 6 and a == eng.generate(ids, frame, max_new_tokens=6, stop_on_eos=False)
    eng.prefill(ids + a[:5], frame)
    via_prefill = eng.logits.float().clone()
    scale = float(via_prefill.abs().max())
    diff = float((via_decode - via_prefill).abs().max())
    assert diff <= 0.03 * scale, f"decode vs prefill logits differ by {diff} (scale {scale})"
    # batched decode at 11B shapes: slot / batch-size invariance (exact) and first token == the single-sequence path
    frame2 = torch.from_numpy(rng.integers(0, 256, (700, 1000, 3), dtype=np.uint8)).to(device)
    ids2 = [1] + rng.integers(1000, cfg.vocab - 8, 90).tolist() + [cfg.image_token_id] + rng.integers(1000, cfg.vocab - 8, 9).tolist()
    ra, rb = (ids, frame), (ids2, frame2)
    out = eng.generate_batch([ra, rb, ra], max_new_tokens=8, stop_on_eos=False)
    assert out[0] == out[2] and out[0][0] == a[0] and [len(t) for t in out] == [8] * 3
    assert eng.generate_batch([rb, ra], max_new_tokens=8, stop_on_eos=False) == [out[1], out[0]]
    del eng
    torch.cuda.empty_cache()


def test_7b_fp8_four_image_batch_at_full_depth(big):
    """BASELINE configs[4]'s per-GPU slice at FULL depth (all 28 / 32 layers, exact 7B shapes): fp8 MFMA prompt pass,
    e4m3 decode weights, four images per step with the text part first (shared prefix).  The oracle comparison of the fp8
    arithmetic is one layer deep (tests/test_fullsize_oracle_gpu.py); here the size-independent properties of the whole
    path: reproducible, slot- and batch-size-invariant, equal with and without the shared prefix.  The comparison with the bf16
    engine at full depth is tests/test_depth_parity_gpu.py::test_7b_full_depth_fp8_vs_bf16 (variance-preserving weights, stated
    bound); on THIS fixture's flat N(0, 0.02) weights every perturbation grows with depth (tools/depth_error.py), so nothing
    is asserted against bf16 here."""
    from vision_inspection_system_amd.engine import Qwen2VLEngine
    cfg, eng, (ra, rb) = big
    dev = eng.device
    eng8 = Qwen2VLEngine(cfg, eng.w, dev, max_ctx=4096, max_batch=4, decode_weights="fp8", prefill_dtype="fp8")
    try:
        rng = np.random.default_rng(21)
        text = rng.integers(0, 1000, 1000).tolist()
        n_img = (980 // 14) ** 2 // 4
        mk = lambda tail: text + [cfg.vision_start_id] + [cfg.image_token_id] * n_img + [cfg.vision_end_id] + tail
        reqs = [(mk([5, 6, 7]), ra[1]), (mk([8, 9, 10]), rb[1]), (mk([5, 6, 7]), ra[1]), (mk([1, 2, 3]), rb[1])]
        out = eng8.generate_batch(reqs, max_new_tokens=10, ignore_eos=True)
        l8 = eng8.logits_b[:4].clone()
        assert len(out) == 4 and all(len(o) == 10 and all(0 <= t < cfg.vocab for t in o) for o in out)
        assert out[0] == out[2] and torch.equal(l8[0], l8[2])                     # slot invariance
        assert eng8.generate_batch(reqs, max_new_tokens=10, ignore_eos=True) == out        # reproducible
        assert eng8.generate_batch([reqs[1], reqs[0]], max_new_tokens=10, ignore_eos=True) == [out[1], out[0]]
        import os
        os.environ["VIS_SHARE_PREFIX"] = "0"
        try:
            assert eng8.generate_batch(reqs, max_new_tokens=10, ignore_eos=True) == out    # shared prefix changes nothing
        finally:
            os.environ.pop("VIS_SHARE_PREFIX")
    finally:
        del eng8
        torch.cuda.empty_cache()
