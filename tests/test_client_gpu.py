"""End-to-end through the drop-in boundary on a real GPU: chat.completions.create -> agents -> nodes -> batch,
provider "mi355x", tiny synthetic model (random weights: the reply text is noise, so the agents must take
their documented failure path without ever raising)."""
import json

import numpy as np
import pytest
from PIL import Image

pytestmark = pytest.mark.gpu


@pytest.fixture
def local_cfg(device):
    from vision_inspection_system_amd import config as C
    cfg = C.Config(vlm_inspector_provider="mi355x", vlm_auditor_provider="mi355x",
                   vlm_inspector_model="synthetic:tiny", vlm_auditor_model="synthetic:tiny:1",
                   vlm_inspector_max_tokens=24, vlm_auditor_max_tokens=16, max_image_dimension=256)
    C.set_config(cfg)
    yield cfg
    C.set_config(None)


@pytest.fixture
def images(tmp_path):
    rng = np.random.default_rng(5)
    out = []
    for i, (h, w) in enumerate([(120, 90), (64, 200), (300, 300)]):
        p = tmp_path / f"img{i}.png"
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(p)
        out.append(str(p))
    return out


def test_chat_completions_shape(local_cfg, images):
    from vision_inspection_system_amd.client import LocalVLMClient, get_model
    from vision_inspection_system_amd.image_processing import encode_image_optimized
    c = LocalVLMClient()
    url = encode_image_optimized(images[0], 256)
    msgs = [{"role": "user", "content": [{"type": "text", "text": "Inspect."},
                                         {"type": "image_url", "image_url": {"url": url}}]}]
    r1 = c.chat.completions.create(model="synthetic:tiny", messages=msgs, temperature=0.0, max_tokens=12)
    r2 = c.chat.completions.create(model="synthetic:tiny", messages=msgs, temperature=0.0, max_tokens=12)
    assert isinstance(r1.choices[0].message.content, str)
    assert r1.choices[0].message.content == r2.choices[0].message.content          # greedy is deterministic
    assert r1.usage["completion_tokens"] <= 12 and r1.usage["prompt_tokens"] > 20
    # temperature > 0: seeded Gumbel-max sampling - reproducible for a fixed seed, a different path than greedy
    s1 = c.chat.completions.create(model="synthetic:tiny", messages=msgs, temperature=1.5, max_tokens=12)
    s2 = c.chat.completions.create(model="synthetic:tiny", messages=msgs, temperature=1.5, max_tokens=12)
    assert s1.choices[0].message.content == s2.choices[0].message.content
    # T -> 0 is the greedy limit (the reference's operating point is T = 0.1 / 0.2: utils/config.py:46-49,:66-69); the
    # distribution itself is checked at kernel level (test_sampling_follows_softmax_of_logits_over_temperature)
    cold = c.chat.completions.create(model="synthetic:tiny", messages=msgs, temperature=1e-6, max_tokens=12)
    assert cold.choices[0].message.content == r1.choices[0].message.content
    # two threads on one engine (Streamlit: one thread per browser session): engine.lock
    # serialises them; both get the single-threaded answer
    import threading
    got, errs = {}, []

    def ask(name, temp):
        try:
            got[name] = [c.chat.completions.create(model="synthetic:tiny", messages=msgs, temperature=temp, max_tokens=12)
                         .choices[0].message.content for _ in range(3)]
        except Exception as e:      # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=ask, args=("greedy", 0.0)), threading.Thread(target=ask, args=("sampled", 1.5))]
    [t.start() for t in th]
    [t.join(120) for t in th]
    assert not errs and not any(t.is_alive() for t in th)
    assert got["greedy"] == [r1.choices[0].message.content] * 3 and got["sampled"] == [s1.choices[0].message.content] * 3
    # text-only health-check style call, no temperature
    r3 = c.chat.completions.create(model="synthetic:tiny", messages=[{"role": "user", "content": "Respond with only the word 'OK'"}], max_tokens=10)
    assert isinstance(r3.choices[0].message.content, str)
    # one engine per (model, device): the agents construct a new client every call
    assert get_model("synthetic:tiny") is get_model("synthetic:tiny")


def test_gpu_resize_path_equals_host_pil_path(local_cfg, images, monkeypatch):
    """Row f3: with the bicubic resample on the GPU (default) the reply is identical to the host-PIL path,
    because the frames are bit-identical."""
    from vision_inspection_system_amd.client import LocalVLMClient
    from vision_inspection_system_amd.image_processing import encode_image_optimized
    c = LocalVLMClient()
    for path in images:
        msgs = [{"role": "user", "content": [{"type": "text", "text": "Inspect."},
                                             {"type": "image_url", "image_url": {"url": encode_image_optimized(path, 256)}}]}]
        monkeypatch.setenv("VIS_GPU_RESIZE", "1")
        a = c.chat.completions.create(model="synthetic:tiny", messages=msgs, temperature=0.0, max_tokens=12)
        monkeypatch.setenv("VIS_GPU_RESIZE", "0")
        b = c.chat.completions.create(model="synthetic:tiny", messages=msgs, temperature=0.0, max_tokens=12)
        assert a.choices[0].message.content == b.choices[0].message.content and a.usage == b.usage


def test_agents_nodes_batch_on_gpu(local_cfg, images):
    from vision_inspection_system_amd.agents import VLMAuditorAgent, VLMInspectorAgent
    from vision_inspection_system_amd.batch import run_batch_inspection
    from vision_inspection_system_amd.schemas import InspectionContext, VLMAnalysisResult
    from vision_inspection_system_amd import nodes
    nodes._sleep = lambda s: None
    ctx = InspectionContext(image_id="a", criticality="medium")
    insp = VLMInspectorAgent()
    res = insp.analyze(images[0], ctx)
    assert isinstance(res, VLMAnalysisResult)          # never raises; random weights -> unparsable reply
    if res.analysis_failed:
        assert "Failed to parse JSON" in res.failure_reason or "Inspector analysis failed" in res.failure_reason
    assert insp.health_check() in (True, False)
    aud = VLMAuditorAgent().verify(images[0], ctx, res)
    assert isinstance(aud, VLMAnalysisResult)
    out = run_batch_inspection(images, "medium", "general")
    assert list(v["image_path"] for v in out["image_results"].values()) == images
    assert out["session_results"]["total_images"] == 3
    for v in out["image_results"].values():
        assert v["completed"] is True and v["safety_verdict"]["verdict"] in ("SAFE", "UNSAFE", "REQUIRES_HUMAN_REVIEW")
        # failed analyses must surface as GATE_0 -> UNSAFE, never as a pass (gates.py:165-184)
        if v["inspector_result"]["analysis_failed"]:
            assert v["safety_verdict"]["verdict"] == "UNSAFE" and "GATE_0_ERROR_STATE" in v["safety_verdict"]["triggered_gates"]
    json.dumps(out, default=str)


def test_mllama_backend_through_the_client_and_auditor(local_cfg, images):
    """Row f2: the Auditor pointed at the mllama family (the reference's HF fallback model) through the same
    chat.completions seam; tiny synthetic weights -> noise text -> documented failure path, never an exception."""
    from vision_inspection_system_amd import config as C
    from vision_inspection_system_amd.agents import VLMAuditorAgent, VLMInspectorAgent
    from vision_inspection_system_amd.client import LocalVLMClient, get_model
    from vision_inspection_system_amd.image_processing import encode_image_optimized
    from vision_inspection_system_amd.schemas import InspectionContext, VLMAnalysisResult
    c = LocalVLMClient()
    url = encode_image_optimized(images[2], 256)
    msgs = [{"role": "user", "content": [{"type": "text", "text": "Verify."},
                                         {"type": "image_url", "image_url": {"url": url}}]}]
    r1 = c.chat.completions.create(model="synthetic:mllama-tiny", messages=msgs, temperature=0.0, max_tokens=10)
    r2 = c.chat.completions.create(model="synthetic:mllama-tiny", messages=msgs, temperature=0.0, max_tokens=10)
    assert r1.choices[0].message.content == r2.choices[0].message.content
    assert r1.usage["completion_tokens"] <= 10 and get_model("synthetic:mllama-tiny").family == "mllama"
    assert r1.timings["sequences"] == 1 and r1.timings["prefill_ms"] > 0 and r1.timings["decode_steps"] >= 0
    r3 = c.chat.completions.create(model="synthetic:mllama-tiny", messages=[{"role": "user", "content": "OK?"}], max_tokens=5)
    assert isinstance(r3.choices[0].message.content, str)
    cfg = C.get_config()
    cfg.vlm_auditor_model = "synthetic:mllama-tiny"
    ctx = InspectionContext(image_id="a", criticality="medium")
    insp = VLMInspectorAgent().analyze(images[0], ctx)
    aud = VLMAuditorAgent().verify(images[0], ctx, insp)
    assert isinstance(aud, VLMAnalysisResult)


def test_batch_inspection_with_mllama_auditor(local_cfg, images):
    """BASELINE configs[2] plumbing: Inspector on the Qwen2-VL engine (batched decode), Auditor on the mllama engine,
    consensus + gates + aggregation, through run_batch_inspection."""
    from vision_inspection_system_amd import config as C
    from vision_inspection_system_amd.batch import run_batch_inspection
    from vision_inspection_system_amd import nodes
    nodes._sleep = lambda s: None                      # no back-off sleeps in tests
    cfg = C.get_config()
    cfg.vlm_auditor_model = "synthetic:mllama-tiny"
    out = run_batch_inspection(images, "high", "aerospace")
    assert list(v["image_path"] for v in out["image_results"].values()) == images
    assert out["session_results"]["total_images"] == 3
    for v in out["image_results"].values():
        assert v["completed"] is True and v["auditor_result"] is not None and v["consensus"] is not None
        assert v["safety_verdict"]["verdict"] in ("SAFE", "UNSAFE", "REQUIRES_HUMAN_REVIEW")


def test_batch_with_an_unreadable_image_streams_the_rest(local_cfg, images, tmp_path, monkeypatch):
    """run_batch_inspection hands the encode futures straight to the local client (agents.prepare_many -> complete_many):
    a file that cannot be opened fails alone (analysis_failed -> GATE_0 -> UNSAFE), the other images are served by the
    shared decode loop, order is kept."""
    from vision_inspection_system_amd import nodes
    from vision_inspection_system_amd.batch import run_batch_inspection
    nodes._sleep = lambda s: None
    bad = tmp_path / "broken.png"
    bad.write_bytes(b"this is not a png")
    paths = [images[0], str(bad), images[1], images[2]]
    monkeypatch.setenv("VIS_MAX_BATCH", "2")      # two groups: the second one's requests are encoded while the first is on the GPU
    out = run_batch_inspection(paths, "medium", "general")
    res = list(out["image_results"].values())
    assert [v["image_path"] for v in res] == paths
    assert res[1]["inspector_result"]["analysis_failed"] is True and res[1]["safety_verdict"]["verdict"] == "UNSAFE"
    assert out["session_results"]["total_images"] == 4 and all(v["completed"] for v in res)


def test_streamed_batch_replies_equal_single_requests(local_cfg, images):
    """complete_many fed with Futures of messages (the streaming form) returns the same replies as one
    chat.completions.create per request (first token exact; greedy continuation up to batched-vs-GEMV near-ties is
    covered at engine level, here the reply of the same request in two places of the batch must be identical)."""
    from vision_inspection_system_amd import ingest
    from vision_inspection_system_amd.agents import VLMInspectorAgent
    agent = VLMInspectorAgent()
    paths = [images[0], images[1], images[0]]

    def request(path):      # a short prompt (the tiny model's context is 1024 byte-tokens), the agent's own image encode
        return [{"role": "user", "content": [{"type": "text", "text": "Inspect this part."},
                                             {"type": "image_url", "image_url": {"url": agent._encode_image_optimized(path)}}]}]

    futs = [ingest.submit(request, p) for p in paths]
    replies = agent.client.complete_many(agent.model_id, futs, 0.0, agent.max_tokens)      # greedy: sampling seeds are per slot
    eager = agent.client.complete_many(agent.model_id, [f.result() for f in futs], 0.0, agent.max_tokens)
    texts = [r.choices[0].message.content for r in replies]
    assert texts == [r.choices[0].message.content for r in eager]
    assert texts[0] == texts[2] and replies[0].usage["prompt_tokens"] == eager[0].usage["prompt_tokens"]
    # per-stage device time rides on the reply (extension): the batch's prompt passes and its shared decode loop
    tm = replies[0].timings
    assert tm["sequences"] == 3 and tm["prefill_ms"] > 0 and tm["decode_ms"] > 0 and tm["decode_steps"] >= 1
    bad = ingest.submit(lambda: (_ for _ in ()).throw(OSError("no such file")))
    mixed = agent.client.complete_many(agent.model_id, [futs[0], bad, futs[1]], 0.0, agent.max_tokens)
    assert isinstance(mixed[1], OSError) and mixed[0].choices[0].message.content == texts[0]


def test_bench_two_ranks_sharing_the_gpu_rehearsal():
    """`bench.py --gpus 2 --share-gpu`: the launcher starts two ranks, each builds its own engine on cuda:0 and inspects its
    own synthetic image, every step's records are exchanged (store vote + gloo all_gather), rank 0 prints the one JSON line.
    Real GPU work through the N > 1 code path on a one-GPU box (RCCL itself cannot put two ranks on one device)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--share-gpu", "--model", "tiny",
                        "--image-size", "112", "--prompt-tokens", "64", "--new-tokens", "8", "--steps", "2", "--warmup", "1",
                        "--no-extras", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["value"] > 0
    assert "REHEARSAL" in out["config"]["parallelism"]


def test_bench_batch256_workload_two_ranks_sharing_the_gpu():
    """`bench.py --gpus 2 --share-gpu --workload batch256` (BASELINE configs[3] as a workload) on tiny weights: rank 0 writes
    the PNG files, both ranks run the whole seam (a3 encode, data-URI decode, engine, parse, consensus, gates) on
    paths[r::2], one gather, strong-scaling JSON with the per-rank engine share."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--share-gpu", "--model", "tiny",
                        "--workload", "batch256", "--images", "10", "--new-tokens", "8"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["value"] > 0 and out["dry"] is False
    assert out["config"]["completed"] == 10 and [p["images"] for p in out["per_rank"]] == [5, 5]
    assert all(p["engine_device_s"] > 0 and p["groups"] >= 1 for p in out["per_rank"]), out["per_rank"]
    assert "REHEARSAL" in out["config"]["parallelism"]


def test_rccl_backend_single_rank_exchange(device):
    """The record exchange on the REAL backend (nccl = RCCL), world size 1 - the most a one-GPU box can run of it: communicator
    init, the store vote, the length + padded-bytes all_gather on device tensors, barrier, MAX all_reduce (bench.py's timing
    reduction).  Runs in a child process: a process group cannot be re-created inside the test runner."""
    import os, subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent(f"""
        import sys, torch, torch.distributed as dist
        sys.path.insert(0, {root!r})
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        from vision_inspection_system_amd import batch
        recs = [{{"image": "a", "tokens": list(range(50))}}, {{"image": "b", "tokens": [1, 2, 3]}}]
        assert batch._all_gather_bytes(dist, b"hello RCCL") == [b"hello RCCL"]
        merged, dead = batch.gather_records_ft(recs, timeout_s=30)
        assert merged == recs and dead == [], (merged, dead)
        assert batch.agree_on("sess", "session_id") == "sess"
        dist.barrier()
        t = torch.tensor([1.5], dtype=torch.float64, device="cuda:0")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t.item()) == 1.5
        dist.destroy_process_group()
        print("RCCL OK")
    """)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "RCCL OK" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])
