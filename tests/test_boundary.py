"""Boundary B1-B4 behaviour with the mock provider (no GPU): client shape, agents never raise,
node state keys / retry / criticality upgrade, batch return shape, config surface."""
import io
import json
import os

import numpy as np
import pytest
from PIL import Image


GOOD_REPLY = json.dumps({
    "object_identified": "steel bracket", "overall_condition": "damaged",
    "defects": [{"type": "Crack", "location": "upper left", "bbox": {"x": 10, "y": 20, "width": 15, "height": 10},
                 "safety_impact": "CRITICAL", "reasoning": "visible fracture", "confidence": "high",
                 "recommended_action": "replace"}],
    "overall_confidence": "high", "analysis_reasoning": "fracture at the weld",
    "inferred_criticality": "high", "inferred_criticality_reasoning": "load bearing"})


@pytest.fixture
def image_path(tmp_path):
    rng = np.random.default_rng(0)
    p = tmp_path / "part.png"
    Image.fromarray(rng.integers(0, 256, (120, 90, 3), dtype=np.uint8)).save(p)
    return p


@pytest.fixture
def mock_cfg(monkeypatch):
    from vision_inspection_system_amd import config as C
    cfg = C.Config(vlm_inspector_provider="mock", vlm_auditor_provider="mock", vlm_inspector_model="m-i",
                   vlm_auditor_model="m-a", max_image_dimension=64)
    C.set_config(cfg)
    yield cfg
    C.set_config(None)


def _patch_reply(monkeypatch, reply):
    from vision_inspection_system_amd import client
    calls = []

    class C(client.CannedResponseClient):
        def __init__(self, **kw):
            super().__init__(reply=reply)
            calls.append(self)
    monkeypatch.setattr(client, "CannedResponseClient", C)
    return calls


def test_client_shape_and_message_format(mock_cfg, image_path, monkeypatch):
    from vision_inspection_system_amd.agents import VLMInspectorAgent
    from vision_inspection_system_amd.schemas import InspectionContext
    made = _patch_reply(monkeypatch, GOOD_REPLY)
    agent = VLMInspectorAgent()
    for attr in ("model_id", "temperature", "max_tokens", "nickname", "is_vision", "logger", "client", "llm"):
        assert hasattr(agent, attr)
    assert agent.nickname == "Inspector" and agent.is_vision and agent.model_id == "m-i"
    res = agent.analyze(image_path, InspectionContext(image_id="x", criticality="low"))
    assert not res.analysis_failed and res.defects[0].type == "crack" and res.critical_defect_count == 1
    call = made[0].calls[0]
    assert call["model"] == "m-i" and call["temperature"] == 0.1 and call["max_tokens"] == 2048
    content = call["messages"][0]["content"]
    assert [p["type"] for p in content] == ["text", "image_url"] and call["messages"][0]["role"] == "user"
    url = content[1]["image_url"]["url"]
    assert url.startswith("data:image/jpeg;base64,")
    # request-side encode: thumbnail to max_image_dimension (64) keeps aspect, JPEG
    from vision_inspection_system_amd.image_processing import decode_data_uri
    assert max(decode_data_uri(url).size) == 64
    assert agent.health_check() is True


def test_agents_never_raise(mock_cfg, image_path, monkeypatch):
    from vision_inspection_system_amd.agents import VLMAuditorAgent, VLMInspectorAgent
    from vision_inspection_system_amd.schemas import InspectionContext, VLMAnalysisResult
    _patch_reply(monkeypatch, "no json here")
    ctx = InspectionContext(image_id="x")
    r = VLMInspectorAgent().analyze(image_path, ctx)
    assert r.analysis_failed and r.failure_reason.startswith("Inspector analysis failed:")
    assert r.object_identified == "unknown" and r.overall_condition == "uncertain" and r.overall_confidence == "low"
    a = VLMAuditorAgent().verify(image_path, ctx, r)
    assert a.analysis_failed and a.failure_reason.startswith("Auditor verification failed:")
    assert a.analysis_reasoning.startswith("Audit verification failed:")
    missing = VLMInspectorAgent().analyze(image_path.parent / "nope.png", ctx)
    assert missing.analysis_failed


def test_inspector_retry_classification(mock_cfg, image_path, monkeypatch):
    """'429'/'rate' -> retried; '413'/'payload' -> ValueError without retry (vlm_inspector.py:113-140)."""
    from vision_inspection_system_amd import agents
    monkeypatch.setattr(agents.time, "sleep", lambda s: None)
    seq = []

    class Flaky:
        class chat:
            class completions:
                @staticmethod
                def create(**kw):
                    seq.append(1)
                    if len(seq) < 3:
                        raise RuntimeError("HTTP 429 too many requests")
                    class M: content = "ok"
                    class Ch: message = M
                    class R: choices = [Ch]
                    return R
    a = agents.VLMInspectorAgent()
    a.client = Flaky
    assert a._call_api_with_retry([{"role": "user", "content": "x"}]) == "ok" and len(seq) == 3

    class Big:
        class chat:
            class completions:
                @staticmethod
                def create(**kw):
                    seq.append(2)
                    raise RuntimeError("413 payload too large")
    a.client = Big
    n = len(seq)
    with pytest.raises(ValueError):
        a._call_api_with_retry([{"role": "user", "content": "x"}])
    assert len(seq) == n + 1


def test_nodes_state_contract(mock_cfg, image_path, monkeypatch):
    from vision_inspection_system_amd import nodes
    _patch_reply(monkeypatch, GOOD_REPLY)
    monkeypatch.setattr(nodes, "_sleep", lambda s: None)
    state = {"image_path": [str(image_path), "ignored.png"], "context": {"image_id": "a", "criticality": "low"},
             "inspector_retry_count": 0}
    out = nodes.run_inspector(state)
    assert out is state and state["current_step"] == "inspector_analysis"
    assert state["inspector_result"]["defects"][0]["type"] == "crack"
    # inferred criticality "high" > user's "low": context upgraded in the dict (nodes.py:188-206)
    assert state["context"]["criticality"] == "high" and state["context"]["criticality_upgraded"] is True
    assert state["context"]["original_criticality"] == "low" and state["context"]["upgrade_reason"] == "load bearing"
    out = nodes.run_auditor(state)
    assert state["current_step"] == "auditor_verification" and not state["auditor_result"]["analysis_failed"]


def test_nodes_failure_path(mock_cfg, image_path, monkeypatch):
    from vision_inspection_system_amd import nodes
    _patch_reply(monkeypatch, "garbage")
    slept = []
    monkeypatch.setattr(nodes, "_sleep", lambda s: slept.append(s))
    state = {"image_path": str(image_path), "context": {"image_id": "a", "criticality": "medium"}}
    nodes.run_inspector(state)
    assert state["inspector_retry_count"] == 1 and slept == [1.0]
    assert state["has_critical_failure"] is True and state["error"].startswith("Inspector failed after 2 attempt(s):")
    assert state["failure_history"] == [state["error"]]
    r = state["inspector_result"]
    assert r["analysis_failed"] and r["failure_reason"] == state["error"]
    assert r["analysis_reasoning"].startswith("Analysis failed after retries:")


def test_batch_return_shape_and_order(mock_cfg, tmp_path, monkeypatch):
    from vision_inspection_system_amd.batch import run_batch_inspection, run_multi_image_inspection
    _patch_reply(monkeypatch, GOOD_REPLY)
    paths = []
    for i in range(3):
        p = tmp_path / f"img{i}.png"
        Image.fromarray(np.full((40, 40, 3), 40 * i, dtype=np.uint8)).save(p)
        paths.append(str(p))
    paths.append(str(tmp_path / "missing.png"))
    id_map = {paths[1]: "custom-id"}
    out = run_multi_image_inspection(paths, criticality="medium", domain="general", session_id="sess", image_id_map=id_map)
    assert set(out) == {"session_id", "image_results", "session_results", "processing_time"}
    ids = list(out["image_results"])
    assert ids[1] == "custom-id" and [out["image_results"][i]["image_path"] for i in ids] == paths
    first = out["image_results"][ids[0]]
    for key in ("inspector_result", "auditor_result", "consensus", "safety_verdict", "clean_verification",
                "explanation", "decision_support", "report_path", "processing_time", "error", "failure_history",
                "completed"):
        assert key in first
    sr = out["session_results"]
    assert sr["total_images"] == 4 and sr["session_id"] == "sess" and "aggregate_verdict" in sr
    assert len(sr["per_image_verdicts"]) == sr["completed_images"]
    json.dumps(out)  # the whole result must be JSON-serialisable (it crosses ranks as JSON)
    out2 = run_batch_inspection(paths[:2], "high", "aerospace")
    assert len(out2["image_results"]) == 2


def test_config_surface(tmp_path, monkeypatch):
    from vision_inspection_system_amd import config as C
    monkeypatch.setenv("VLM_INSPECTOR_MODEL", "/models/qwen2-vl-7b")
    monkeypatch.setenv("VLM_INSPECTOR_PROVIDER", "mi355x")
    monkeypatch.setenv("VLM_INSPECTOR_MAX_TOKENS", "777")
    cfg = C.Config.from_env()
    assert (cfg.vlm_inspector_model, cfg.vlm_inspector_provider, cfg.vlm_inspector_max_tokens) == \
        ("/models/qwen2-vl-7b", "mi355x", 777)
    assert cfg.vlm_auditor_temperature == 0.2 and cfg.max_image_dimension == 2048
    y = tmp_path / "models.yaml"
    y.write_text("inspector:\n  model_id: synthetic:tiny\n  temperature: 0.0\n  max_tokens: 64\n  description: d\n"
                 "  provider: mi355x\nauditor:\n  model_id: x/y\n  temperature: 0.2\n  max_tokens: 1024\n"
                 "  provider: huggingface\ngroq:\n  enabled: false\n")
    cfg.apply_models_yaml(str(y))
    assert cfg.vlm_inspector_model == "synthetic:tiny" and cfg.vlm_inspector_max_tokens == 64
    assert cfg.vlm_auditor_model == "x/y" and cfg.vlm_auditor_provider == "huggingface"
    # the reference's shipped models.yaml parses with the same loader (schema check; file content is the app's)
    ref = "/root/reference/config/models.yaml"
    if os.path.exists(ref):
        blocks = C.load_models_yaml(ref)
        assert set(blocks) == {"inspector", "auditor", "explainer"} and blocks["inspector"].provider == "huggingface"


def test_local_client_refuses_unknown_model_without_gpu_side_effects():
    from vision_inspection_system_amd.client import LocalVLMClient, resolve_model_dir
    assert resolve_model_dir("Qwen/Qwen2-VL-7B-Instruct") is None
    c = LocalVLMClient()
    with pytest.raises(Exception) as ei:
        c.chat.completions.create(model="Qwen/Qwen2-VL-7B-Instruct", messages=[{"role": "user", "content": "hi"}],
                                  max_tokens=4)
    msg = str(ei.value).lower()
    for needle in ("429", "rate", "413", "payload"):
        assert needle not in msg, f"error text must not trip the reference's retry classifier: {msg}"


def test_chat_layout_and_tokenizer():
    from vision_inspection_system_amd.config import Qwen2VLConfig
    from vision_inspection_system_amd.tokenizer import ByteTokenizer, build_chat_ids
    cfg = Qwen2VLConfig.tiny()
    tok = ByteTokenizer(cfg.vocab, cfg.image_token_id, cfg.vision_start_id, cfg.vision_end_id, cfg.eos_ids)
    msgs = [{"role": "user", "content": [{"type": "text", "text": "Hi"},
                                         {"type": "image_url", "image_url": {"url": "data:..."}}]}]
    ids = build_chat_ids(tok, msgs, [6])
    sys_part = [256] + list(b"system\nYou are a helpful assistant.") + [503] + list(b"\n")
    assert ids[:len(sys_part)] == sys_part
    user = [256] + list(b"user\nHi") + [501] + [500] * 6 + [502, 503] + list(b"\n")
    assert ids[len(sys_part):len(sys_part) + len(user)] == user
    assert ids[-len(b"assistant\n") - 1:] == [256] + list(b"assistant\n")
    assert tok.decode(list(b"ok") + [503]) == "ok"
    with pytest.raises(ValueError):
        build_chat_ids(tok, msgs, [])


def test_unsupported_model_type_is_refused_up_front(tmp_path):
    """The reference's code default is Qwen/Qwen2.5-VL-7B-Instruct (utils/config.py:42-45); a local directory of a
    family the engines do not implement must fail with a clear message, not a KeyError inside a weight loader."""
    from vision_inspection_system_amd import client
    d = tmp_path / "some-model"
    d.mkdir()
    (d / "config.json").write_text(json.dumps({"model_type": "llava_next", "text_config": {}}))
    with pytest.raises(ValueError, match="model_type 'llava_next'"):
        client.get_model(str(d), device="cuda:0")
    (d / "config.json").unlink()
    with pytest.raises(FileNotFoundError, match="config.json"):
        client.get_model(str(d), device="cuda:0")
    with pytest.raises(FileNotFoundError, match="not a local directory"):
        client.get_model("Qwen/Qwen2.5-VL-7B-Instruct", device="cuda:0")


def test_plumbing_baseline_runs_without_gpu():
    """BASELINE configs[0]: the canned-client batch run bench.py reports as ``plumbing``."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    out = bench.plumbing_baseline(n_images=4, size=64)
    assert out["images"] == 4 and out["completed"] == 4 and out["value"] > 0 and out["unit"] == "images/s"


def test_batched_agent_calls_use_the_ingest_pool_and_keep_failures_per_image(mock_cfg, tmp_path, monkeypatch):
    """analyze_many / verify_many encode on the ingest thread pool (ingest.py); order is preserved, an unreadable image is
    a per-image analysis_failed result, prepared futures (prepare_many) are accepted."""
    from vision_inspection_system_amd import ingest
    from vision_inspection_system_amd.agents import VLMAuditorAgent, VLMInspectorAgent
    from vision_inspection_system_amd.schemas import InspectionContext
    _patch_reply(monkeypatch, GOOD_REPLY)
    paths = []
    for i in range(5):
        p = tmp_path / f"im{i}.png"
        if i != 2:
            Image.fromarray(np.full((50 + i, 40, 3), 30 * i, dtype=np.uint8)).save(p)
        paths.append(p)
    ctxs = [InspectionContext(image_id=f"i{i}", criticality="medium") for i in range(5)]
    assert ingest.threads() >= 1
    for agent, call in ((VLMInspectorAgent(), "analyze_many"), (VLMAuditorAgent(), "verify_many")):
        res = getattr(agent, call)(paths, ctxs)
        assert [r.analysis_failed for r in res] == [False, False, True, False, False]
        assert "im2.png" in res[2].failure_reason or "No such file" in res[2].failure_reason
        futs = agent.prepare_many(paths, ctxs)
        res2 = getattr(agent, call)(paths, ctxs, prepared=futs)
        assert [r.analysis_failed for r in res2] == [False, False, True, False, False]
        assert res2[0].defects[0].type == "crack"


def test_encode_is_shared_between_the_agents_when_the_bytes_are_the_same(tmp_path, monkeypatch):
    """Inspector and Auditor encode the same file; for an image within the Auditor's 1024 px limit that is not in mode LA the
    bytes are identical, so the second request takes the first one's result (image_processing encode cache).  A different
    thumbnail decision or conversion is a different key; a rewritten file is a different key; the 10 MB refusal is the
    caller's own (applied on the cached path too)."""
    from vision_inspection_system_amd import image_processing as IP
    IP.clear_encode_cache()
    calls = []
    real = IP._encode
    monkeypatch.setattr(IP, "_encode", lambda *a: (calls.append(a[1:3]), real(*a))[1])
    rng = np.random.default_rng(3)
    small, big, la = tmp_path / "s.png", tmp_path / "b.png", tmp_path / "la.png"
    Image.fromarray(rng.integers(0, 256, (300, 200, 3), dtype=np.uint8)).save(small)
    Image.fromarray(rng.integers(0, 256, (1200, 300, 3), dtype=np.uint8)).save(big)
    Image.fromarray(rng.integers(0, 256, (60, 40, 2), dtype=np.uint8), mode="LA").save(la)
    insp = dict(max_size=2048, convert_la=True, enforce_limit=True)
    aud = dict(max_size=1024, convert_la=False, enforce_limit=False)
    a, b = IP.encode_image_optimized(small, **insp), IP.encode_image_optimized(small, **aud)
    assert a == b and len(calls) == 1                       # shared
    assert IP.encode_image_optimized(big, **insp) != IP.encode_image_optimized(big, **aud) and len(calls) == 3   # 1200 > 1024
    IP.encode_image_optimized(la, **insp)
    with pytest.raises(OSError):                            # PIL cannot write LA as JPEG: the Auditor's own (reference) behaviour
        IP.encode_image_optimized(la, **aud)
    assert len(calls) == 5
    # same path, new content -> new key
    import os
    Image.fromarray(rng.integers(0, 256, (300, 200, 3), dtype=np.uint8)).save(small)
    os.utime(small, ns=(1, 1))
    c = IP.encode_image_optimized(small, **insp)
    assert c != a and len(calls) == 6
    # the size refusal is per caller, also when the bytes come from the cache
    IP.clear_encode_cache()
    monkeypatch.setattr(IP, "_encode", lambda *a: ("data:image/jpeg;base64,AAAA", 10_000_001))
    assert IP.encode_image_optimized(small, **aud).endswith("AAAA")
    with pytest.raises(ValueError, match="too large"):
        IP.encode_image_optimized(small, **insp)
    # switched off
    monkeypatch.setenv("VIS_ENCODE_CACHE_MB", "0")
    n = []
    monkeypatch.setattr(IP, "_encode", lambda *a: (n.append(1), ("x", 1))[1])
    IP.encode_image_optimized(small, **aud); IP.encode_image_optimized(small, **aud)
    assert len(n) == 2
    IP.clear_encode_cache()


def test_streaming_client_gets_the_encode_futures_and_failures_stay_per_image(mock_cfg, tmp_path):
    """A client that sets accepts_futures (LocalVLMClient) is handed the prepare_many futures themselves and answers a
    failed request with the exception in its place; the agent turns exactly that one into analysis_failed."""
    from concurrent.futures import Future
    from vision_inspection_system_amd.agents import VLMInspectorAgent
    from vision_inspection_system_amd.schemas import InspectionContext
    paths = []
    for i in range(4):
        p = tmp_path / f"im{i}.png"
        if i != 1:
            Image.fromarray(np.full((40, 40, 3), 50 * i, dtype=np.uint8)).save(p)
        paths.append(p)
    seen = {}

    class StreamingClient:
        accepts_futures = True

        def complete_many(self, model, batch, temperature=None, max_tokens=None):
            seen["futures"] = all(isinstance(m, Future) for m in batch)
            out = []
            for j, m in enumerate(batch):
                try:
                    m.result()
                except Exception as e:
                    out.append(e)
                    continue
                if j == 3:
                    out.append(ValueError("decode failed"))
                    continue
                out.append(type("R", (), {"choices": [type("C", (), {"message": type("M", (), {"content": GOOD_REPLY})()})()]})())
            return out

    agent = VLMInspectorAgent()
    agent.client = StreamingClient()
    res = agent.analyze_many(paths, [InspectionContext(image_id=f"i{i}", criticality="medium") for i in range(4)])
    assert seen["futures"] is True
    assert [r.analysis_failed for r in res] == [False, True, False, True]
    assert "decode failed" in res[3].failure_reason


def test_direct_frames_skip_the_jpeg_round_trip_only_on_request(tmp_path, monkeypatch):
    """VIS_DIRECT_FRAMES=1 (off by default: the reference always sends JPEG q85): the local agents hand the a3-prepared
    RGB pixels over under a process-local vis-frame: URL; the service side resolves it to exactly those pixels."""
    from vision_inspection_system_amd import config as C, image_processing as IP
    from vision_inspection_system_amd.agents import VLMAuditorAgent, VLMInspectorAgent
    rng = np.random.default_rng(9)
    p = tmp_path / "big.png"
    Image.fromarray(rng.integers(0, 256, (1500, 700, 4), dtype=np.uint8), mode="RGBA").save(p)
    C.set_config(C.Config(vlm_inspector_provider="mi355x", vlm_auditor_provider="mi355x", vlm_inspector_model="synthetic:tiny",
                          vlm_auditor_model="synthetic:tiny:1"))
    try:
        insp, aud = VLMInspectorAgent(), VLMAuditorAgent()
        assert insp._encode_image_optimized(p).startswith("data:image/jpeg;base64,")          # default: the reference's bytes
        monkeypatch.setenv("VIS_DIRECT_FRAMES", "1")
        for agent, max_size in ((insp, 2048), (aud, 1024)):
            url = agent._encode_image_optimized(p)
            assert url.startswith("vis-frame:")
            ref = Image.open(p)
            if max(ref.size) > max_size:
                ref.thumbnail((max_size, max_size), Image.Resampling.LANCZOS)
            assert np.array_equal(np.array(IP.decode_data_uri(url)), np.array(ref.convert("RGB")))
        with pytest.raises(ValueError, match="unknown or expired"):
            IP.decode_data_uri("vis-frame:999999999")
        # a remote provider never gets a process-local handle
        C.set_config(C.Config(vlm_inspector_provider="mock", vlm_auditor_provider="mock"))
        assert VLMAuditorAgent()._encode_image_optimized(p).startswith("data:image/jpeg")
    finally:
        C.set_config(None)


def test_ingest_pool_priorities_and_chaining(monkeypatch):
    """ingest.then queues the dependent (decode) task only when its input is ready and ahead of the (encode) tasks still
    waiting - no worker ever blocks on another task; failures and a shutdown travel through the chained future."""
    import threading, time
    from vision_inspection_system_amd import ingest
    monkeypatch.setenv("VIS_INGEST_THREADS", "2")
    ingest.shutdown()
    try:
        order, gate = [], threading.Event()

        def enc(i):
            gate.wait(5)
            order.append(("enc", i))
            return i

        def dec(i):
            order.append(("dec", i))
            return i * 10

        encs = [ingest.submit(enc, i) for i in range(6)]
        decs = [ingest.then(f, dec) for f in encs]
        gate.set()
        assert [d.result(10) for d in decs] == [0, 10, 20, 30, 40, 50]
        # a decode never waits behind all encodes: dec 0 runs before the last encode
        assert order.index(("dec", 0)) < order.index(("enc", 5))
        assert ingest.then(7, dec).result(10) == 70                       # plain value: just a task
        bad = ingest.submit(lambda: 1 / 0)
        ran = []
        chained = ingest.then(bad, lambda v: ran.append(v))
        with pytest.raises(ZeroDivisionError):
            chained.result(10)
        assert ran == []
        with pytest.raises(KeyError):
            ingest.then(ingest.submit(lambda: 3), lambda v: {}[v]).result(10)
        # shutdown: queued work is cancelled, its dependants fail instead of hanging
        hold = threading.Event()
        blockers = [ingest.submit(hold.wait, 5) for _ in range(2)]
        time.sleep(0.05)
        queued = ingest.submit(lambda: 1)
        dep = ingest.then(queued, lambda v: v)
        ingest.shutdown()
        hold.set()
        with pytest.raises(BaseException):
            dep.result(10)
        assert ingest.submit(lambda: 5).result(10) == 5                   # a fresh pool is created on demand
    finally:
        ingest.shutdown()


def test_direct_frame_handles_end_with_their_request(tmp_path, monkeypatch):
    """VIS_DIRECT_FRAMES handles are released when the request that carried them is over (a frame is ~3 MB and
    nothing else frees it), and the backstop cap follows VIS_MAX_BATCH: two groups x two agents + retries must fit."""
    import numpy as np
    from PIL import Image
    from vision_inspection_system_amd import image_processing as IP
    p = tmp_path / "f.png"
    Image.fromarray(np.zeros((40, 50, 3), np.uint8)).save(p)
    urls = [IP.frame_url_for(p) for _ in range(3)]
    msgs = [[{"role": "user", "content": [{"type": "text", "text": "x"}, {"type": "image_url", "image_url": {"url": u}}]}]
            for u in urls]
    assert IP.decode_data_uri(urls[0]).size == (50, 40)
    IP.release_frames(msgs[0])
    IP.release_frames(None)
    IP.release_frames([{"role": "user", "content": "text only"}])
    with pytest.raises(ValueError):
        IP.decode_data_uri(urls[0])
    assert IP.decode_data_uri(urls[1]).size == (50, 40)          # the others are untouched
    monkeypatch.setenv("VIS_MAX_BATCH", "64")
    assert IP.frame_cap() == 512
    monkeypatch.setenv("VIS_MAX_BATCH", "8")
    assert IP.frame_cap() == 256
    IP.release_frames(msgs[1]); IP.release_frames(msgs[2])
