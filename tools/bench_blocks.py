"""Measured blocks that ride on bench.py's default line next to the configs[1] headline: the
per-GPU slices of BASELINE configs[2] / [3] / [4], each timed with its own HIP events and carrying its own roofline
figure, so that the batch numbers are driver-timed instead of builder-run.

  batch64     configs[3] per-GPU slice, kernel-only: 64 x 1024^2 images, text part first (the reference's message order,
              src/agents/vlm_inspector.py:462-470), per-image prompt pass + ONE shared 128-token decode loop
  fp8_batch4  configs[4] per-GPU slice: 4 images per GPU, fp8 MFMA prompt pass, e4m3 decode weights
  dual        configs[2]: one image through Qwen2-VL-7B + Llama-3.2-11B-Vision + consensus + safety gates, and the same
              for 32 images per step (both models' decode loops batched)
  seam64      the whole seam on 64 PNG files: run_batch_inspection -> agents (a3 encode) -> client (JPEG decode, resize) ->
              engine -> parse -> consensus -> gates -> aggregate, Inspector on the local engine, Auditor canned

The unit these replace is the reference's sequential per-image loop, src/orchestration/graph.py:308-357.  Weights are
seeded random at the exact shapes (no checkpoint offline): timing is valid, replies are noise."""
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _bench():
    import bench           # the root script: helpers only (its main() is guarded)
    return bench


def batch_block(engine, frame, n_patches, n_img_tok, B, new, prompt_tokens, steps=2):
    """Kernel-only batch step on ``engine`` (max_batch >= B): B prompt passes (shared text prefix) + (new - 1) replays of
    the B-sequence decode graph; HIP events around both parts; the weight-streaming projection kernel replayed alone
    for its roofline figure."""
    bn = _bench()
    cfg = engine.cfg
    ids = bn.synthetic_prompt(cfg, n_img_tok, prompt_tokens, order="text-first")
    S = len(ids)
    ids_dev = torch.tensor(ids, dtype=torch.int32, device=engine.device)

    def step():
        s, m, e = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        s.record()
        engine.prefill_many([(ids, [frame])] * B, max_new_tokens=new, ids_dev=[ids_dev] * B)
        m.record()
        g = engine._ensure_graph(B)
        for _ in range(new - 1):
            g.replay()
        e.record()
        toks = engine.tokens_b[:B, S - 1:S - 1 + new].cpu()      # D2H of the B x new ids (synchronises)
        return s, m, e, toks

    step()                                                        # warm-up: graph capture, caches
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    evs = [step() for _ in range(steps)]
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps
    t_pre = sum(s.elapsed_time(m) for s, m, e, _ in evs) / steps * 1e-3
    t_dec = sum(m.elapsed_time(e) for s, m, e, _ in evs) / steps * 1e-3
    fp8 = engine.decode_weights == "fp8" and engine.fp8_batched
    k_avg, k_launches = bn.measure_decode_gemm(engine, B)
    step_bytes = bn.gemv_bytes_per_step(cfg) / (2 if fp8 else 1)
    if fp8:                                                       # the o projection stays bf16 in the batched fp8 step
        step_bytes += cfg.layers * cfg.heads * cfg.head_dim * cfg.hidden
    bpl = step_bytes / k_launches
    flops = bn.prefill_flops(cfg, n_patches, S) * B
    P = engine.shared_prefix_len([ids] * B) if (B > 1 and os.environ.get("VIS_SHARE_PREFIX", "1") != "0") else 0
    flops_ex = bn.prefill_flops_executed(cfg, n_patches, S, P, B) * B
    peak_tf = bn.MFMA_BF16_PEAK_TF * (2 if engine.prefill_dtype == "fp8" else 1)
    traffic, traffic_src = (bn.measured_traffic("decode_stream", True) if (B == 64 and not fp8) else
                            bn.measured_traffic("decode_stream_fp8", True) if (B == 4 and fp8) else (None, None))
    return {
        "images_per_s": B / wall, "ms_per_step": wall * 1e3, "batch": B, "steps": steps, "prompt_tokens": S,
        "new_tokens": new, "prompt_pass_ms_per_image": t_pre * 1e3 / B, "decode_ms_per_step": t_dec * 1e3 / (new - 1),
        "prefill_mfma": {"achieved": flops / t_pre / 1e12, "peak": peak_tf, "unit": "TFLOP/s",
                         "frac": flops / t_pre / 1e12 / peak_tf,
                         "executed_achieved": flops_ex / t_pre / 1e12, "executed_frac": flops_ex / t_pre / 1e12 / peak_tf,
                         "shared_prefix_tokens": P,
                         "note": "achieved / frac: FLOPs of B full prompt passes / time of the B passes (the shared text prefix "
                                 "runs once per batch, so this is an EFFECTIVE rate); executed_*: only the arithmetic that ran "
                                 "(vision tower + suffix rows + the prefix once per batch)"
                                 + ("; LLM + ViT projections on the fp8 MFMA, attention bf16: priced against the fp8 dense peak"
                                    if engine.prefill_dtype == "fp8" else "")},
        "roofline": {"bound": "hbm", "kernel": ("decode_proj_kernel" if engine.fused_proj else "gemm_decode_stream_kernel")
                     + ("<fp8>" if fp8 else ""),
                     "achieved": bpl / k_avg / 1e9, "peak": bn.HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": bpl / k_avg / 1e9 / bn.HBM_PEAK_GBS,
                     # PMC passes exist for the bf16 kernel at 64 rows and for the fp8 kernel at 4 rows (profiles/*_traffic.json)
                     "traffic": traffic, "traffic_source": traffic_src,
                     "bytes_per_launch": bpl, "avg_launch_us": k_avg * 1e6, "launches_per_step": k_launches},
        "dtype": ("fp8-e4m3 prompt-pass projections / " if engine.prefill_dtype == "fp8" else "bf16 prompt pass / ")
        + ("fp8-e4m3 decode weights" if engine.decode_weights == "fp8" else "bf16 decode weights"),
    }


def mllama_prefill_flops(mc, n_rows: int, n_present: int, S: int) -> float:
    """Arithmetic of one Llama-3.2-11B-Vision prompt pass: the vision tower over the n_rows-token tile canvas (32 local + 8
    global layers, attention inside the present tiles' tokens), the projector, the 32 self-attention + 8 cross-attention
    decoder layers over S prompt rows (cross keys / values projected from the n_present vision tokens), lm_head of one row."""
    E, H, D = mc.v_hidden, mc.hidden, mc.head_dim
    vl = mc.v_layers + mc.v_global_layers
    vit = 2.0 * n_rows * (3 * mc.patch * mc.patch * E + vl * (4 * E * E + 2 * E * mc.v_mlp)) + vl * 4.0 * n_present * n_present * E
    vit += 2.0 * n_present * mc.v_out * H
    n_cross = len(mc.cross_layers)
    n_self = mc.layers - n_cross
    kvw = mc.kv_heads * D
    mlp = 3 * H * mc.intermediate
    self_l = 2.0 * S * (H * (H + 2 * kvw) + H * H + mlp) + 2.0 * S * S * mc.heads * D
    cross_l = 2.0 * S * (2 * H * H + mlp) + 2.0 * n_present * 2 * H * kvw + 4.0 * S * n_present * mc.heads * D
    return vit + n_self * self_l + n_cross * cross_l + 2.0 * H * mc.vocab


def dual_block(insp, aud, B, new, prompt_tokens=700, steps=2):
    """configs[2]: B images per step through the Inspector (Qwen2-VL-7B) and the Auditor (Llama-3.2-11B-Vision), replicas
    resident together, then the reference's post-processing per image (consensus + safety gates on the two results;
    random weights reply noise, which parses to the documented failure object - same host work)."""
    from vision_inspection_system_amd import hip
    from vision_inspection_system_amd.consensus import analyze_consensus
    from vision_inspection_system_amd.gates import evaluate_safety
    from vision_inspection_system_amd.image_processing import smart_resize
    from vision_inspection_system_amd.schemas import InspectionContext, VLMAnalysisResult
    qc, mc, dev = insp.cfg, aud.cfg, insp.device
    rng = np.random.default_rng(0)
    raw = torch.from_numpy(rng.integers(0, 256, (1024, 1024, 3), dtype=np.uint8)).to(dev)
    th, tw = smart_resize(1024, 1024)
    n_img = (th // qc.patch) * (tw // qc.patch) // qc.merge ** 2
    q_text = rng.integers(0, 1000, prompt_tokens).tolist()
    q_ids = q_text + [qc.vision_start_id] + [qc.image_token_id] * n_img + [qc.vision_end_id] + [5, 6]   # text part first
    m_ids = [1] + rng.integers(1000, mc.vocab - 8, prompt_tokens).tolist() + [mc.image_token_id, 5, 6]
    ctx = InspectionContext(image_id="bench", criticality="medium")
    failed = VLMAnalysisResult(object_identified="unknown", overall_condition="uncertain", defects=[], overall_confidence="low",
                               analysis_failed=True, failure_reason="Failed to parse JSON")
    rows, verdict = [], None
    for it in range(steps + 1):
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        t0 = time.perf_counter()
        ev[0].record()
        frame = hip.resize_rgb(raw, th, tw)
        if B > 1:
            insp.generate_batch([(q_ids, [frame])] * B, max_new_tokens=new, ignore_eos=True)
            ev[1].record()
            aud.generate_batch([(m_ids, raw)] * B, max_new_tokens=new, stop_on_eos=False)
        else:
            insp.prefill(q_ids, [frame], max_new_tokens=new)
            insp.decode(new - 1)
            ev[1].record()
            aud.prefill(m_ids, raw)
            aud.decode(new - 1)
            insp.generated(new), aud.generated(new)
        ev[2].record()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(B):
            verdict = evaluate_safety(analyze_consensus(failed, failed), ctx)
        t2 = time.perf_counter()
        if it:
            rows.append((t1 - t0, t2 - t1, ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])))
    gpu, host, t_i, t_a = (float(np.mean([r[i] for r in rows])) for i in range(4))
    out = {"batch": B, "images_per_s": B / (gpu + host), "ms_per_image": (gpu + host) * 1e3 / B,
           "inspector_ms_per_image": t_i / B, "auditor_ms_per_image": t_a / B, "postprocess_ms_per_image": host * 1e3 / B,
           "new_tokens_per_model": new, "verdict": verdict.verdict, "resident_GB": torch.cuda.memory_allocated() / 1e9}
    if B > 1:
        # roofline of the step's two halves: the prompt passes against the dense bf16 MFMA peak (executed arithmetic: the
        # Inspector's text prefix and the Auditor's identical prompt run once per group, not per image - see the notes), the
        # shared decode loops against HBM (weights streamed once per step for all B sequences)
        bn = _bench()
        ti, ta = dict(insp.last_timing), dict(aud.last_timing)
        n_patches = (th // qc.patch) * (tw // qc.patch)
        Sq = len(q_ids)
        P = insp.shared_prefix_len([q_ids] * B)
        f_i = bn.prefill_flops_executed(qc, n_patches, Sq, P, B)
        f_a = mllama_prefill_flops(mc, aud.TP if hasattr(aud, "TP") else 6432, 4 * mc.tile_tokens, len(m_ids))
        wb_i = bn.gemv_bytes_per_step(qc)
        wb_a = sum(t.numel() * 2 for lw in aud.w.layers for t in (lw.qkv_w, lw.o_w, lw.gateup_w, lw.down_w) if t is not None) \
            + aud.w.lm_head.numel() * 2
        steps_i, steps_a = max(1, ti.get("decode_steps", new - 1)), max(1, ta.get("decode_steps", new - 1))
        out["roofline"] = {
            "prompt_pass": {"bound": "mfma", "peak": bn.MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
                            "inspector": {"ms_per_image": ti["prefill_ms"] / B, "executed_TF_per_image": f_i / 1e12,
                                          "achieved": f_i * B / (ti["prefill_ms"] * 1e-3) / 1e12,
                                          "frac": f_i * B / (ti["prefill_ms"] * 1e-3) / 1e12 / bn.MFMA_BF16_PEAK_TF},
                            "auditor": {"ms_per_image": ta["prefill_ms"] / B, "executed_TF_per_image": f_a / 1e12,
                                        "achieved": f_a * B / (ta["prefill_ms"] * 1e-3) / 1e12,
                                        "frac": f_a * B / (ta["prefill_ms"] * 1e-3) / 1e12 / bn.MFMA_BF16_PEAK_TF,
                                        "note": "per-image arithmetic of the tower + decoder; the text rows of a group's identical "
                                                "prompts are stacked, not skipped"}},
            "decode": {"bound": "hbm", "peak": bn.HBM_PEAK_GBS, "unit": "GB/s",
                       "inspector": {"ms_per_step": ti["decode_ms"] / steps_i, "weight_bytes_per_step": wb_i,
                                     "achieved": wb_i / (ti["decode_ms"] / steps_i * 1e-3) / 1e9,
                                     "frac": wb_i / (ti["decode_ms"] / steps_i * 1e-3) / 1e9 / bn.HBM_PEAK_GBS},
                       "auditor": {"ms_per_step": ta["decode_ms"] / steps_a, "weight_bytes_per_step": wb_a,
                                   "achieved": wb_a / (ta["decode_ms"] / steps_a * 1e-3) / 1e9,
                                   "frac": wb_a / (ta["decode_ms"] / steps_a * 1e-3) / 1e9 / bn.HBM_PEAK_GBS},
                       "note": "whole decode step (projections + attention over the KV caches + finalisations) against the weight "
                               "bytes alone; kernels per step: profiles/r04_dual_kernel_stats.csv"}}
    return out


def seam_block(engine, n_images=64, size=1024, new=128, model_id="synthetic:bench-seam"):
    """run_batch_inspection on ``n_images`` PNG files through every layer of the seam, Inspector on ``engine`` (registered
    under ``model_id``), Auditor on the canned-response client, one rank.  Random weights reply noise; VIS_SYNTHETIC_REPLY
    substitutes a parseable reply AFTER the full generation so that the agents stay on their success path."""
    from PIL import Image
    from vision_inspection_system_amd import client as CL, config as C, ingest
    from vision_inspection_system_amd.batch import run_batch_inspection
    from vision_inspection_system_amd.image_processing import clear_encode_cache
    from vision_inspection_system_amd.tokenizer import ByteTokenizer
    cfg = engine.cfg
    reply = ('{"object_identified": "part", "overall_condition": "good", "defects": [], "overall_confidence": "high", '
             '"analysis_reasoning": "no visible damage"}')
    saved_env = {k: os.environ.get(k) for k in ("VIS_IGNORE_EOS", "VIS_SYNTHETIC_REPLY", "VIS_MAX_BATCH")}
    os.environ.update(VIS_IGNORE_EOS="1", VIS_SYNTHETIC_REPLY=reply, VIS_MAX_BATCH=str(engine.max_batch))
    old_cfg = C.get_config()
    tok = ByteTokenizer(cfg.vocab, cfg.image_token_id, cfg.vision_start_id, cfg.vision_end_id, cfg.eos_ids)
    CL.register_model(model_id, str(engine.device), CL.LoadedModel(engine, tok, cfg, model_id))
    CL.set_mock_reply(reply)
    C.set_config(C.Config(vlm_inspector_provider="mi355x", vlm_inspector_model=model_id, vlm_inspector_max_tokens=new,
                          vlm_inspector_temperature=0.0, vlm_auditor_provider="mock", vlm_auditor_model="mock",
                          vlm_auditor_max_tokens=new, vlm_auditor_temperature=0.0))
    try:
        with tempfile.TemporaryDirectory() as d:
            t0 = time.perf_counter()
            paths = [os.path.join(d, f"frame{i:03d}.png") for i in range(n_images)]

            def write(i):     # seeded uint8 frames, written as PNG so that the a3 encode does its JPEG round trip
                rng = np.random.default_rng(1234 + i)
                Image.fromarray(rng.integers(0, 256, (size, size, 3), dtype=np.uint8)).save(paths[i], compress_level=1)
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 4)) as pool:     # (zlib releases the GIL: untimed set-up)
                list(pool.map(write, range(n_images)))
            t_files = time.perf_counter() - t0
            run_batch_inspection(paths[:4], "medium", "general")          # warm: graphs for this token budget, pool threads
            ingest.shutdown()
            clear_encode_cache()                                           # the measured run encodes its own images
            torch.cuda.synchronize()
            ingest.TRACE = []
            from vision_inspection_system_amd import hip as _hip
            _hip.call_trace_start()
            t0 = time.perf_counter()
            out = run_batch_inspection(paths, "medium", "general")
            torch.cuda.synchronize()
            t = time.perf_counter() - t0
            calls = _hip.call_trace_stop()
            trace, ingest.TRACE = ingest.TRACE, None
        timing = dict(getattr(engine, "last_timing", {}))
        stages = {}
        for name, a, b, th in trace:       # per stage: calls, busy seconds (summed over threads), first start / last end
            st = stages.setdefault(name, {"calls": 0, "busy_s": 0.0, "first_start_s": 1e9, "last_end_s": 0.0, "threads": set()})
            st["calls"] += 1
            st["busy_s"] += b - a
            st["first_start_s"] = min(st["first_start_s"], a - t0)
            st["last_end_s"] = max(st["last_end_s"], b - t0)
            st["threads"].add(th)
        for st in stages.values():
            st["threads"] = len(st["threads"])
        return {"images_per_s": n_images / t, "seconds": t, "images": n_images, "image_px": size,
                "completed": out["session_results"]["completed_images"], "new_tokens": new,
                "ingest_threads": int(os.environ.get("VIS_INGEST_THREADS", "4")), "host_cpus": os.cpu_count(),
                "engine_device_ms": {"prompt_passes": timing.get("prefill_ms"), "decode_loop": timing.get("decode_ms")},
                "host_timeline": stages,
                # the thread that launches the kernels: entry-point calls of the measured call and the time spent inside them
                # (argument marshalling + hipLaunchKernel; the GIL is released during the C call) - against `seconds`
                "launch_thread": (lambda c: {"library_calls": c[0], "inside_library_s": c[1]})(
                    max(calls.values(), key=lambda c: c[0]) if calls else [0, 0.0]),
                "write_png_files_s": t_files,
                "what": "run_batch_inspection on PNG files: a3 encode (PIL thumbnail / JPEG q85 / base64) -> data-URI decode "
                        "(host Huffman + GPU IDCT) -> GPU resize -> tokenise -> per-image prompt pass + shared decode loop -> "
                        "detokenise -> parse -> validate -> consensus -> gates -> aggregate; Inspector local, Auditor canned; "
                        "engine_device_ms includes the engine waiting for lazily decoded requests"}
    finally:
        CL.set_mock_reply(None)
        CL.unregister_model(model_id, str(engine.device))
        C.set_config(old_cfg)
        for k, v in saved_env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def run_all(cfg, weights, dev, frame, n_patches, n_img_tok, new, prompt_tokens, log=lambda s: None):
    """All blocks, each isolated: a failing block reports its error string and the others still run."""
    from vision_inspection_system_amd import mllama_weights as MW
    from vision_inspection_system_amd.engine import Qwen2VLEngine
    from vision_inspection_system_amd.mllama_engine import MllamaEngine
    import logging
    out = {}
    # the post-processing of the dual / seam blocks warns once per image (noise replies disagree): not on a timing run
    quiet = [logging.getLogger(n) for n in ("vision_inspection_system_amd", "agent")]
    levels = [lg.level for lg in quiet]
    for lg in quiet:
        lg.setLevel(logging.ERROR)

    def guarded(name, fn):
        t0 = time.perf_counter()
        try:
            out[name] = fn()
        except Exception as e:      # noqa: BLE001 - an extra block must never take the headline down
            out[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
        out[name]["block_wall_s"] = time.perf_counter() - t0
        log(f"{name}: {time.perf_counter() - t0:.1f} s")

    eng64 = None
    try:
        eng64 = Qwen2VLEngine(cfg, weights, dev, max_ctx=4096, max_batch=64)
    except Exception as e:      # noqa: BLE001
        out["batch64"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    if eng64 is not None:
        guarded("batch64", lambda: batch_block(eng64, frame, n_patches, n_img_tok, 64, new, prompt_tokens))
        guarded("seam64", lambda: seam_block(eng64, 64, 1024, new))
        if "error" not in out["seam64"] and "error" not in out["batch64"]:
            out["seam64"]["kernel_only_images_per_s"] = out["batch64"]["images_per_s"]

        def n1():     # BASELINE configs[3] on ONE GPU: the measured N = 1 point of the strong-scaling curve
            r = seam_block(eng64, 256, 1024, new)
            r["workload"] = ("configs[3] at N = 1: run_batch_inspection over 256 x 1024x1024 PNG files on ONE GPU (four decode "
                             "batches of 64), Inspector local, Auditor canned - the same call `bench.py --gpus 8 --workload "
                             "batch256` shards over ranks; no 8-GPU run exists")
            r.pop("host_timeline", None)
            return r
        guarded("batch256_n1", n1)

        def dual():
            mc = MW.MllamaConfig.mllama_11b()
            aud = MllamaEngine(mc, MW.random_device_weights(mc, dev, 1), dev, max_ctx=2048, max_batch=32)
            try:
                return {"workload": "configs[2]: Qwen2-VL-7B Inspector + Llama-3.2-11B-Vision Auditor on one GPU, 1024x1024, "
                                    f"{new}+{new} greedy tokens, consensus + safety gates",
                        "single": dual_block(eng64, aud, 1, new), "batch32": dual_block(eng64, aud, 32, new)}
            finally:
                del aud
        guarded("dual", dual)
    del eng64
    torch.cuda.empty_cache()

    def fp8():
        eng = Qwen2VLEngine(cfg, weights, dev, max_ctx=4096, max_batch=4, decode_weights="fp8", prefill_dtype="fp8")
        try:
            return batch_block(eng, frame, n_patches, n_img_tok, 4, new, prompt_tokens, steps=3)
        finally:
            del eng
    guarded("fp8_batch4", fp8)
    torch.cuda.empty_cache()
    for lg, lv in zip(quiet, levels):
        lg.setLevel(lv)
    return out
