#!/usr/bin/env python3
"""Headline benchmark: inspected images/sec, Qwen2-VL-7B, synthetic 1024x1024 frames (BASELINE.json).

A "step" is one pass of the hot path over one image: K1 patchify -> ViT -> merger -> LLM prefill over
S = 2249 tokens (1225 image + 1024 text) -> exactly 128 greedy tokens (EOS ignored), the configuration
BASELINE.json's metric is quoted on (configs[1]).  Inputs (resized u8 frame, token ids) are resident in
HBM before the timed region.  Weights are seeded random bf16 at the exact 7B shapes (no checkpoint exists
offline), so timing is valid and the generated text is noise.

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL.  Either the caller starts the ranks (``python -m torch.distributed.run
--nproc-per-node N ... bench.py --gpus N``: RANK / WORLD_SIZE are then in the environment) or - when
WORLD_SIZE is unset - this script starts them ITSELF before it has touched the GPU (a child
``torch.distributed.run`` process; the parent only relays rank 0's JSON line and the exit status).  Whole
images are sharded across ranks (weak scaling: every rank inspects its own image stream), each step ends with
one RCCL all_gather of the per-image result records, and the reported time is the max over ranks.
``--backend gloo --dry-device cpu`` runs the same launcher / rendezvous / gather / max-over-ranks plumbing
with NO model (records only) - the CPU rehearsal of the N > 1 path used by tests/test_bench_launcher.py.

The single JSON line also carries
  roofline     - the dominant kernel (decode weight-streaming GEMV, HBM-bound): algorithmic bytes per
                 launch / average launch duration, measured live with HIP events;
  prefill_mfma - prefill FLOPs / prefill time against the dense bf16 MFMA peak;
  cpu_baseline - the oracle (CPU port of the same arithmetic) timed on this host's cores on a bounded
                 sample of the same workload;
  microbench   - measured peak-GEMM and stream-copy figures of THIS box with the
                 fractions of the prefill / decode rates against them;
  e2e          - what one ``chat.completions.create`` costs on the same 1024x1024 JPEG (data-URI decode, GPU
                 resize, tokenise, prefill, decode, detokenise) next to the kernel-only step;
  plumbing     - BASELINE configs[0]: run_multi_image_inspection with a canned-response client on 448x448
                 frames (host logic only, no GPU), images/s;
  batch64, seam64, dual, fp8_batch4 - the per-GPU slices of configs[3] / [2] / [4] (tools/bench_blocks.py), each
                 with its own HIP events and roofline figure; the headline value / config stay configs[1].
"""
import argparse
import io
import json
import os
import sys
import time

# multi-process GPU work on this pool needs dmabuf IPC (RCCL fails with the legacy mode); harmless for one process
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s measured achievable)
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16 MFMA peak


def synthetic_frame(i: int, size: int) -> np.ndarray:
    """Seeded uniform-random RGB frame -> reference-style JPEG q85 round trip (a3) -> smart_resize u8 frame."""
    from PIL import Image
    from vision_inspection_system_amd.image_processing import resize_for_model
    rng = np.random.default_rng(1234 + i)
    img = Image.fromarray(rng.integers(0, 256, (size, size, 3), dtype=np.uint8))
    buf = io.BytesIO()
    img.save(buf, format="JPEG", quality=85, optimize=True)
    buf.seek(0)
    return resize_for_model(Image.open(buf).convert("RGB"))


def synthetic_prompt(cfg, n_image_tokens: int, n_text: int, seed: int = 99, order: str = "image-first"):
    """n_text text ids in total.  image-first: [text_a(16) | <vision_start> | image pads | <vision_end> | text_b];
    text-first (the reference's part order, vlm_inspector.py:462-470): [text_a | <vision_start> | ... | <vision_end> |
    text_b(8)] - the inspection prompt precedes the image, so the images of a batch share it."""
    rng = np.random.default_rng(seed)
    hi = min(151643, cfg.vocab - 16)
    text = rng.integers(0, hi, n_text - 2).tolist()
    cut = 16 if order == "image-first" else len(text) - 8
    return text[:cut] + [cfg.vision_start_id] + [cfg.image_token_id] * n_image_tokens + [cfg.vision_end_id] + text[cut:]


def prefill_flops(cfg, n_patches: int, S: int) -> float:
    E, M = cfg.v_embed, cfg.v_embed * cfg.merge ** 2
    v25 = getattr(cfg, "vision_arch", "qwen2_vl") == "qwen2_5_vl"
    mlp_mats = 3 if v25 else 2                                     # SwiGLU: gate, up, down
    vit_gemm = 2.0 * n_patches * (cfg.patch_dim * E + cfg.v_depth * (3 * E * E + E * E + mlp_mats * E * cfg.v_mlp))
    vit_gemm += 2.0 * (n_patches // cfg.merge ** 2) * (M * M + M * cfg.hidden)
    vit_attn = cfg.v_depth * 4.0 * n_patches * n_patches * E
    if v25:   # only the v_fullatt blocks attend over the whole image; the others inside windows of <= (v_window/patch)^2 patches
        win = (cfg.v_window // cfg.patch) ** 2
        n_full = len([i for i in cfg.v_fullatt if i < cfg.v_depth])
        vit_attn = 4.0 * n_patches * E * (n_full * n_patches + (cfg.v_depth - n_full) * win)
    D = cfg.head_dim
    per_layer = cfg.hidden * (cfg.heads + 2 * cfg.kv_heads) * D + cfg.heads * D * cfg.hidden \
        + 3 * cfg.hidden * cfg.intermediate
    llm_gemm = 2.0 * S * cfg.layers * per_layer + 2.0 * cfg.hidden * cfg.vocab
    llm_attn = cfg.layers * 2.0 * S * S * cfg.heads * D  # causal: half of 4*S^2*H*D
    return vit_gemm + vit_attn + llm_gemm + llm_attn


def prefill_flops_executed(cfg, n_patches: int, S: int, P: int, B: int) -> float:
    """FLOPs actually executed per image when B images share a text prefix of P tokens (computed once per batch): the whole
    vision tower, the LLM projections over the S - P suffix rows + 1 / B of the prefix rows, causal attention of those rows."""
    full = prefill_flops(cfg, n_patches, S)
    if P <= 0 or B <= 1:
        return full
    D = cfg.head_dim
    per_layer = cfg.hidden * (cfg.heads + 2 * cfg.kv_heads) * D + cfg.heads * D * cfg.hidden + 3 * cfg.hidden * cfg.intermediate
    llm_gemm_full = 2.0 * S * cfg.layers * per_layer
    llm_attn_full = cfg.layers * 2.0 * S * S * cfg.heads * D
    rows = (S - P) + P / B
    sq = (S * S - P * P) + P * P / B
    return full - llm_gemm_full - llm_attn_full + 2.0 * rows * cfg.layers * per_layer + cfg.layers * 2.0 * sq * cfg.heads * D


def gemv_bytes_per_step(cfg) -> float:
    D = cfg.head_dim
    per_layer = cfg.hidden * (cfg.heads + 2 * cfg.kv_heads) * D + cfg.heads * D * cfg.hidden \
        + 3 * cfg.hidden * cfg.intermediate
    return 2.0 * (cfg.layers * per_layer + cfg.hidden * cfg.vocab)


def measure_gemv(engine, reps: int = 5):
    """Average duration of one gemv_bf16_kernel launch, measured live with HIP events on the launch stream.

    The GEMV launches of one decode step (same weights, same buffers, same arguments as the real step: 113 in the
    unchained step; 56 - gate/up and down - when the head of every layer runs as the chained launch: its qkv / o
    projections are not this kernel, and the lm_head then runs as the kernel's argmax-epilogue instance) are captured alone into a hipGraph and the replay is bracketed by two events, so the figure is
    (sum of kernel durations + in-graph kernel boundaries) / launches.  Bracketing every tiny kernel with its
    own event pair instead adds ~5 us of event overhead per launch and over-reads by ~20 %; rocprofv3's
    per-kernel average (profiles/) is the cross-check."""
    from vision_inspection_system_amd import hip
    cfg, w = engine.cfg, engine.w
    n = [0]

    def gemvs():
        x, x2 = engine.d_x, engine.d_x2
        if engine.decode_weights == "fp8":
            for lw, q in zip(w.llm, engine.q8):
                hip.gemv_fp8(x[0], *q["qkv_w"], engine.d_qkv, bias=lw.qkv_b, norm_w=lw.ln1_w, eps=cfg.rms_eps)
                hip.gemv_fp8(engine.d_attn, *q["o_w"], x2[0], residual=x[0])
                hip.gemv_fp8(x2[0], *q["gateup_w"], engine.d_act, norm_w=lw.ln2_w, act=hip.ACT_SWIGLU, eps=cfg.rms_eps)
                hip.gemv_fp8(engine.d_act, *q["down_w"], x[0], residual=x2[0])
                n[0] += 4
            hip.gemv_fp8(x[0], *engine.q8_lm_head, engine.logits, norm_w=w.final_norm_w, eps=cfg.rms_eps)
            n[0] += 1
            return
        chained = engine.chain_sync is not None
        for lw in w.llm:
            if not chained:
                hip.gemv(x[0], lw.qkv_w, engine.d_qkv, bias=lw.qkv_b, norm_w=lw.ln1_w, eps=cfg.rms_eps)
                hip.gemv(engine.d_attn, lw.o_w, x2[0], residual=x[0])
                n[0] += 2
            hip.gemv(x2[0], lw.gateup_w, engine.d_act, norm_w=lw.ln2_w, act=hip.ACT_SWIGLU, eps=cfg.rms_eps)
            hip.gemv(engine.d_act, lw.down_w, x[0], residual=x2[0])
            n[0] += 2
        if not chained:     # (chained step: the lm_head runs as the kernel's argmax-epilogue instance, a row of its own in rocprof)
            hip.gemv(x[0], w.lm_head, engine.logits, norm_w=w.final_norm_w, eps=cfg.rms_eps)
            n[0] += 1

    side = torch.cuda.Stream(device=engine.device)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        gemvs()
    torch.cuda.current_stream().wait_stream(side)
    launches = n[0]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        gemvs()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e-3 / reps / launches, launches


def measured_traffic(kind: str = "gemv", with_source: bool = False):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/*_<kind>_traffic.json: "gemv" = gemv_bf16_kernel, "decode_stream" = the batched decode projection kernel), or
    None.  The figure is NOT measured in this run (PMC passes need their own rocprofv3 invocations): ``with_source`` also
    returns the tracked file it was read from, which the JSON line names next to it ("traffic_source")."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_{kind}_traffic.json")))
    val, src = None, None
    if files:
        try:
            with open(files[-1]) as f:
                val, src = float(json.load(f)["hbm_bytes_per_launch"]), "profiles/" + os.path.basename(files[-1])
        except Exception:
            val, src = None, None
    return (val, src) if with_source else val


def cpu_baseline(cfg, n_patches: int, S: int, new_tokens: int):
    """Oracle (CPU port) on a bounded sample: one ViT block, one LLM prefill layer, decode layers + lm_head,
    at the exact 7B shapes, fp32, all host threads; extrapolated to one full image."""
    from oracle import qwen2vl_ref as R
    torch.manual_seed(0)
    rc = R.RefConfig(hidden=cfg.hidden, layers=1, heads=cfg.heads, kv_heads=cfg.kv_heads,
                     intermediate=cfg.intermediate, vocab=cfg.vocab, v_depth=1, v_embed=cfg.v_embed,
                     v_heads=cfg.v_heads, v_mlp=cfg.v_mlp, image_token_id=cfg.image_token_id)
    E, H, D = cfg.v_embed, cfg.hidden, cfg.head_dim

    def rn(*s):
        return torch.randn(*s) * 0.02

    sd = {"visual.patch_embed.proj.weight": rn(E, 3, 2, 14, 14)}
    p = "visual.blocks.0."
    for n_, s_ in (("norm1.weight", (E,)), ("norm1.bias", (E,)), ("norm2.weight", (E,)), ("norm2.bias", (E,)),
                   ("attn.qkv.weight", (3 * E, E)), ("attn.qkv.bias", (3 * E,)), ("attn.proj.weight", (E, E)),
                   ("attn.proj.bias", (E,)), ("mlp.fc1.weight", (cfg.v_mlp, E)), ("mlp.fc1.bias", (cfg.v_mlp,)),
                   ("mlp.fc2.weight", (E, cfg.v_mlp)), ("mlp.fc2.bias", (E,))):
        sd[p + n_] = rn(*s_)
    M = E * 4
    for n_, s_ in (("visual.merger.ln_q.weight", (E,)), ("visual.merger.ln_q.bias", (E,)),
                   ("visual.merger.mlp.0.weight", (M, M)), ("visual.merger.mlp.0.bias", (M,)),
                   ("visual.merger.mlp.2.weight", (H, M)), ("visual.merger.mlp.2.bias", (H,))):
        sd[n_] = rn(*s_)
    p = "model.layers.0."
    for n_, s_ in (("input_layernorm.weight", (H,)), ("post_attention_layernorm.weight", (H,)),
                   ("self_attn.q_proj.weight", (cfg.heads * D, H)), ("self_attn.q_proj.bias", (cfg.heads * D,)),
                   ("self_attn.k_proj.weight", (cfg.kv_heads * D, H)), ("self_attn.k_proj.bias", (cfg.kv_heads * D,)),
                   ("self_attn.v_proj.weight", (cfg.kv_heads * D, H)), ("self_attn.v_proj.bias", (cfg.kv_heads * D,)),
                   ("self_attn.o_proj.weight", (H, cfg.heads * D)), ("mlp.gate_proj.weight", (cfg.intermediate, H)),
                   ("mlp.up_proj.weight", (cfg.intermediate, H)), ("mlp.down_proj.weight", (H, cfg.intermediate))):
        sd[p + n_] = rn(*s_)
    sd["model.norm.weight"] = torch.ones(H)
    lm_head = rn(cfg.vocab, H)
    side = int(round(n_patches ** 0.5))
    with torch.no_grad():
        # every piece runs twice and the SECOND timing is kept (the first pays thread-pool start-up and page faults of the
        # multi-GB operands: an un-warmed single sample moved 272 -> 344 -> 400 s per image across boxes)
        pv = torch.randn(n_patches, cfg.patch_dim)
        for _ in range(2):
            t0 = time.perf_counter()
            R.vision_forward(rc, sd, pv, [(1, side, side)])
            t_vit = time.perf_counter() - t0       # patch embed + 1 block + merger
        x = torch.randn(S, H)
        cos, sin = R.mrope_cos_sin(rc, torch.arange(S).view(1, -1).expand(3, -1))
        for _ in range(2):
            cache = R.KVCache(1)
            t0 = time.perf_counter()
            R.text_forward(rc, sd, x, cos, sin, cache)
            t_llm = time.perf_counter() - t0       # 1 prefill layer
        c1, s1 = R.mrope_cos_sin(rc, torch.full((3, 1), S))
        n_dec = 4
        h = R.text_forward(rc, sd, torch.randn(1, H), c1, s1, cache)      # warm
        t0 = time.perf_counter()
        for _ in range(n_dec):
            h = R.text_forward(rc, sd, torch.randn(1, H), c1, s1, cache)
        t_dec = (time.perf_counter() - t0) / n_dec  # 1 decode layer
        _ = h[-1] @ lm_head.t()                     # warm
        t0 = time.perf_counter()
        for _ in range(2):
            _ = h[-1] @ lm_head.t()
        t_lm = (time.perf_counter() - t0) / 2
    t_img = cfg.v_depth * t_vit + cfg.layers * t_llm + new_tokens * (cfg.layers * t_dec + t_lm)
    host = "unknown"
    try:   # quote the baseline with the box it ran on (the lease lands on different hosts: never a ratio across runs)
        with open("/proc/cpuinfo") as f:
            host = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "unknown")
    except OSError:
        pass
    return {"value": 1.0 / t_img, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "host_cpu": f"{host} ({os.cpu_count()} logical CPUs)",
            "sample": (f"oracle fp32 at 7B shapes: 1 ViT block N={n_patches} ({t_vit:.2f}s), 1 LLM prefill layer "
                       f"S={S} ({t_llm:.2f}s), {n_dec} decode-layer steps ({t_dec * 1e3:.0f}ms each), lm_head "
                       f"({t_lm * 1e3:.0f}ms); extrapolated x{cfg.v_depth}/x{cfg.layers}/x{new_tokens} to one image "
                       f"= {t_img:.1f}s; every piece warmed once, second timing kept. The reference has no CPU arithmetic path (remote API).")}


def measure_decode_gemm(engine, B: int, reps: int = 5):
    """Batched decode (--batch B): average duration of one gemm_decode_stream_kernel launch - the 113 weight-streaming
    projections of one step for B sequences, same weights and buffers as the real step, captured alone into a
    hipGraph and replayed between two HIP events (the finalisation / attention / sampling kernels are left out)."""
    from vision_inspection_system_amd import hip
    cfg, w = engine.cfg, engine.w
    fp8 = engine.decode_weights == "fp8" and engine.fp8_batched
    n = [0]

    def projections():
        if engine.fused_proj:        # r05: decode_proj_kernel (projection + reduction + epilogue in one launch)
            n[0] += engine._decode_step_fused(B, projections_only=True)
            return
        xn, xn2, att, act, part = engine.b_xn[:B], engine.b_xn2[:B], engine.b_attn[:B], engine.b_act[:B], engine.b_part
        if fp8:
            xq, x2q, aq = engine.b_xq[:B], engine.b_x2q[:B], engine.b_actq[:B]
            s0, s1, s2 = engine.b_sx[0, :B], engine.b_sx[1, :B], engine.b_sx[2, :B]
            for lw, q8 in zip(w.llm, engine.q8):
                hip.decode_gemm_fp8(xq, s0, *q8["qkv_w"], part=part)
                hip.decode_gemm(att, lw.o_w, part=part)
                hip.decode_gemm_fp8(x2q, s1, *q8["gateup_w"], part=part)
                hip.decode_gemm_fp8(aq, s2, *q8.get("down_w_pad", q8["down_w"]), part=part)
                n[0] += 4
            hip.decode_gemm_fp8(xq, s0, *engine.q8_lm_head, out=engine.logits_b[:B])
        else:
            for lw in w.llm:
                hip.decode_gemm(xn, lw.qkv_w, part=part)
                hip.decode_gemm(att, lw.o_w, part=part)
                hip.decode_gemm(xn2, lw.gateup_w, part=part)
                hip.decode_gemm(act, lw.down_w, part=part)
                n[0] += 4
            hip.decode_gemm(xn, w.lm_head, out=engine.logits_b[:B])
        n[0] += 1

    side = torch.cuda.Stream(device=engine.device)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        projections()
    torch.cuda.current_stream().wait_stream(side)
    launches = n[0]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        projections()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e-3 / reps / launches, launches


def microbench(dev):
    """What THIS box delivers, next to the spec-sheet peaks the roofline fractions are quoted against (BASELINE.md
    section 3): a dense bf16 GEMM at 8192^3 on N(0,1) operands - the vendor library through torch.matmul (hipBLASLt;
    an independent yardstick, not on the product path) and this repository's own tile kernel - and a 1 GiB
    device-to-device stream copy (read + write bytes)."""
    from vision_inspection_system_amd import hip
    out = {}
    n = 8192
    a = torch.randn((n, n), dtype=torch.bfloat16, device=dev)
    b = torch.randn((n, n), dtype=torch.bfloat16, device=dev)
    c = torch.empty((n, n), dtype=torch.bfloat16, device=dev)

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) * 1e-3 / reps

    flops = 2.0 * n ** 3
    try:
        t = timed(lambda: torch.matmul(a, b.t(), out=c), 10)
        out["gemm_bf16_8192_library_TFLOPs"] = flops / t / 1e12
    except Exception as ex:     # the yardstick must never take the benchmark down
        out["gemm_bf16_8192_library_TFLOPs"] = None
        out["library_error"] = str(ex)[:120]
    t = timed(lambda: hip.gemm(a, b, out=c), 10)
    out["gemm_bf16_8192_own_kernel_TFLOPs"] = flops / t / 1e12
    del a, b, c
    src = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    dst = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    src.fill_(1)
    t = timed(lambda: dst.copy_(src), 10)
    out["stream_copy_GBps"] = 2.0 * (1 << 30) / t / 1e9
    # a read-only sweep (the decode GEMV only reads): the vendor reduction over the same 1 GiB - an independent
    # yardstick for a read stream; a copy both reads and writes and is not a ceiling for it
    try:
        v = src.view(torch.float32)
        t = timed(lambda: v.sum(), 10)
        out["stream_read_GBps"] = (1 << 30) / t / 1e9
    except Exception as ex:
        out["stream_read_GBps"] = None
        out["read_error"] = str(ex)[:120]
    out["note"] = ("8192^3 bf16 on N(0,1) operands; copy = 1 GiB device-to-device, read + write bytes; read = torch.sum over "
                   "1 GiB of float32 (vendor reduction kernel, read-only stream)")
    return out


def e2e_request(engine, cfg, image_size: int, prompt_tokens: int, new: int, reps: int = 3):
    """One ``LocalVLMClient.chat.completions.create`` on the same synthetic frame, the way an agent calls it: a
    base64 JPEG data URI (a3's output) + a text part -> decode JPEG (host libjpeg), upload, GPU bicubic resize,
    tokenise, prefill, ``new`` decode steps (EOS ignored so that the work equals the kernel-only step), detokenise."""
    import base64
    from PIL import Image
    from vision_inspection_system_amd import client as CL
    from vision_inspection_system_amd.tokenizer import ByteTokenizer
    rng = np.random.default_rng(1234)
    img = Image.fromarray(rng.integers(0, 256, (image_size, image_size, 3), dtype=np.uint8))
    buf = io.BytesIO()
    img.save(buf, format="JPEG", quality=85, optimize=True)
    uri = "data:image/jpeg;base64," + base64.b64encode(buf.getvalue()).decode()
    tok = ByteTokenizer(cfg.vocab, cfg.image_token_id, cfg.vision_start_id, cfg.vision_end_id, cfg.eos_ids)
    model_id = "synthetic:bench-engine"
    CL.register_model(model_id, str(engine.device), CL.LoadedModel(engine, tok, cfg, model_id))
    text = ("inspect this part for defects and answer in JSON. " * 64)[:max(16, prompt_tokens - 64)]  # 1 byte = 1 token
    messages = [{"role": "user", "content": [{"type": "text", "text": text},
                                             {"type": "image_url", "image_url": {"url": uri}}]}]
    cl = CL.LocalVLMClient(device=str(engine.device))
    os.environ["VIS_IGNORE_EOS"] = "1"
    try:
        times, usage = [], None
        for i in range(reps + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = cl.chat.completions.create(model=model_id, messages=messages, temperature=0.0, max_tokens=new)
            torch.cuda.synchronize()
            if i:
                times.append(time.perf_counter() - t0)
            usage = r.usage
        # cold form: the text prefix is NOT in the engine's prefix cache (a first request with this prompt); graphs and
        # position tables stay warm, as they would be for any earlier request of another prompt
        engine._prefix_cache.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cl.chat.completions.create(model=model_id, messages=messages, temperature=0.0, max_tokens=new)
        torch.cuda.synchronize()
        cold = time.perf_counter() - t0
    finally:
        os.environ.pop("VIS_IGNORE_EOS", None)
        CL.unregister_model(model_id, str(engine.device))
    t = sum(times) / len(times)
    return {"ms": t * 1e3, "images_per_s": 1.0 / t, "cold_prefix_ms": cold * 1e3, "jpeg_bytes": len(buf.getvalue()),
            "prompt_tokens": usage["prompt_tokens"], "completion_tokens": usage["completion_tokens"],
            "what": "LocalVLMClient.chat.completions.create: base64 + JPEG Huffman decode (host) + H2D + GPU IDCT / colour + GPU "
                    "bicubic resize + tokenise + prefill + decode + detokenise, one request at a time; text part first as the "
                    "reference sends it, so from the second request on the prompt's text prefix comes from the engine's prefix "
                    "cache (the kernel-only step recomputes it every time)"}


GOOD_REPLY = ('```json\n{"object_identified": "steel bracket", "overall_condition": "damaged", "defects": [{"type": '
              '"crack", "location": "upper left weld seam", "bbox": {"x": 12.5, "y": 20.0, "width": 18.0, "height": 9.5},'
              ' "safety_impact": "CRITICAL", "reasoning": "a dark linear discontinuity crosses the weld bead", '
              '"confidence": "high", "recommended_action": "replace"}], "overall_confidence": "high", '
              '"analysis_reasoning": "one clear crack at the weld"}\n```')


def plumbing_baseline(n_images: int = 24, size: int = 448):
    """BASELINE configs[0] / SURVEY section 8(d)(i): the host side of the path with NO model - the counterpart of
    run_multi_image_inspection (src/orchestration/graph.py:269-387) driven by a canned-response client: PNG open ->
    thumbnail/JPEG q85/base64 (a3) x 2 agents -> parse (a6) -> validate (a7) -> pydantic (a8) -> consensus -> gates ->
    aggregate, on seeded uniform-random frames; one process, images/s."""
    import tempfile
    from PIL import Image
    from vision_inspection_system_amd import client as CL, config as C
    from vision_inspection_system_amd.batch import run_multi_image_inspection
    old = C.get_config()
    CL.set_mock_reply(GOOD_REPLY)
    C.set_config(C.Config(vlm_inspector_provider="mock", vlm_auditor_provider="mock"))
    try:
        with tempfile.TemporaryDirectory() as d:
            paths = []
            for i in range(n_images):
                rng = np.random.default_rng(1234 + i)
                path = os.path.join(d, f"frame{i:03d}.png")
                Image.fromarray(rng.integers(0, 256, (size, size, 3), dtype=np.uint8)).save(path)
                paths.append(path)
            run_multi_image_inspection(paths[:2])
            t0 = time.perf_counter()
            out = run_multi_image_inspection(paths, criticality="medium", domain="general")
            t = time.perf_counter() - t0
        done = out["session_results"]["completed_images"]
    finally:
        CL.set_mock_reply(None)
        C.set_config(old)
    return {"value": n_images / t, "unit": "images/s", "images": n_images, "completed": done, "image_px": size,
            "cores": 1, "host_cpus": os.cpu_count(),
            "what": "configs[0]: run_multi_image_inspection, canned-response client, no model, no GPU (host logic only)"}


def free_port() -> int:
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n: int, argv) -> int:
    """``python bench.py --gpus N`` without a launcher around it: start N ranks as ONE child process group
    (torch.distributed.run, 127.0.0.1 rendezvous) and relay their output.  Called before this process has made any
    GPU call - nothing here touches torch.cuda, and the GPU work happens only in the children, so no process that
    has initialised the GPU is ever replaced.  Returns the exit status (non-zero if any rank failed or rank 0 printed
    no JSON line)."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    got_json = False
    for line in proc.stdout:
        if line.startswith("{") and '"metric"' in line:
            got_json = True
        sys.stdout.write(line)
        sys.stdout.flush()
    rc = proc.wait()
    if rc == 0 and not got_json:
        print("bench.py: the ranks exited cleanly but rank 0 printed no result line", file=sys.stderr)
        return 1
    return rc


def run_dry(args, rank: int, world: int) -> None:
    """CPU rehearsal of the N > 1 plumbing: rendezvous, per-step record exchange, barriers, max over ranks, rank 0's
    JSON line - with synthetic token records instead of a model.  The figure it prints measures nothing."""
    import torch.distributed as dist
    from vision_inspection_system_amd.batch import gather_records
    if world > 1:
        dist.init_process_group(args.backend)
    rng = np.random.default_rng(rank)

    def one_step():
        toks = rng.integers(0, 1000, args.new_tokens).tolist()
        rec = gather_records([{"image": f"synthetic_{rank}_{b}", "tokens": toks} for b in range(args.batch)], world)
        if len(rec) != world * args.batch:
            raise RuntimeError(f"rank {rank}: gathered {len(rec)} records, expected {world * args.batch}")
    for _ in range(args.warmup):
        one_step()
    if world > 1:
        _cpu_barrier(dist)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    if world > 1:
        _cpu_barrier(dist)
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        print(json.dumps({"metric": "DRY RUN (no model): record exchange only", "value": world * args.steps * args.batch / elapsed,
                          "unit": "records/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "none", "data": "synthetic", "dry": True,
                          "config": {"workload": "launcher / gather rehearsal on CPU", "backend": args.backend,
                                     "parallelism": f"dp{world}"}}), flush=True)
    if world > 1:
        _cpu_barrier(dist)
        dist.destroy_process_group()


def _cpu_barrier(dist) -> None:
    """Barrier of a CPU-only (gloo) rehearsal: an all_reduce of one CPU element.  dist.barrier() asks PyTorch for the current
    accelerator and thereby opens the GPU in every rank even when it is hidden - a dry run must not touch it."""
    dist.all_reduce(torch.zeros(1))


class HostOnlyEngine:
    """--dry-ingest: stands where the Inspector's engine stands and consumes its requests WITHOUT a model or a GPU, so that
    run_batch_inspection does everything the seam does on the HOST - a3 encode (PIL thumbnail / JPEG q85 / base64) and the
    service-side base64 + Huffman decode on the ingest pool, tokenisation, reply parsing, consensus, gates, aggregation - and
    nothing else.  What it times is the host ceiling of a rank (the 8-GPU expectation assumes eight ranks'
    ingest pools and launch threads do not contend).  Test / measurement infrastructure, never selected by the product."""
    host_only = True

    def __init__(self, cfg, max_batch: int):
        import threading
        self.cfg, self.max_batch, self.device = cfg, max_batch, "cpu"
        self.lock = threading.Lock()
        self.last_timing: dict = {}

    def generate_batch(self, requests, max_new_tokens: int = 128, **_kw) -> list:
        out = []
        for r in requests:
            try:
                ids, _frames = r() if callable(r) else r        # waits for the request's encode + Huffman decode
                out.append([65, 66, 67, 68])
            except Exception as e:      # noqa: BLE001 - a failed request keeps its exception, as with the real engine
                out.append(e)
        return out


def run_batch256(args, rank: int, world: int, local_rank: int) -> None:
    """BASELINE configs[3] as a workload: the unit is the reference's batch call,
    run_multi_image_inspection (src/orchestration/graph.py:269-387; README run_batch_inspection), on ``--images`` PNG files.
    Rank 0 writes the files, the directory is agreed through the rendezvous store, every rank calls run_batch_inspection
    on the FULL list (rank r inspects paths[r::W]; the records move with ONE all_gather pair) - so the time of the call,
    max over ranks, is the strong-scaling time of the batch.  Each rank runs the whole seam: a3 encode on its ingest pool
    (VIS_INGEST_THREADS sized from the host cores per rank), data-URI decode, engine, parse, consensus, gates.
    ``--dry-device cpu``: both agents on the canned-response client (no model, no GPU): launcher, file agreement,
    sharding, gather and timing only - the CPU rehearsal used by tests/test_bench_launcher.py."""
    import tempfile
    import torch.distributed as dist
    from PIL import Image
    dry = args.dry_device == "cpu"
    if not dry:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a ROCm GPU: the hot path has no CPU fallback")
        if args.share_gpu:
            local_rank = 0
        torch.cuda.set_device(local_rank)
    if world > 1:
        from datetime import timedelta
        if dry or args.share_gpu:
            dist.init_process_group("gloo", timeout=timedelta(seconds=900))
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=timedelta(seconds=900))
    from vision_inspection_system_amd import client as CL, config as C
    from vision_inspection_system_amd.batch import agree_on, gather_records, run_batch_inspection
    from vision_inspection_system_amd.image_processing import clear_encode_cache
    cores = os.cpu_count() or world
    os.environ.setdefault("VIS_INGEST_THREADS", str(max(1, min(4, cores // world - 1))))   # more threads only fight the launch thread for the GIL (8: -15 %)
    os.environ["VIS_IGNORE_EOS"] = "1"
    reply = ('{"object_identified": "part", "overall_condition": "good", "defects": [], "overall_confidence": "high", '
             '"analysis_reasoning": "no visible damage"}')
    os.environ["VIS_SYNTHETIC_REPLY"] = reply           # random weights answer noise: keep the agents on the success path
    CL.set_mock_reply(reply)
    model_id = {"7b": "synthetic:7b", "7b25": "synthetic:qwen2.5-vl-7b", "tiny": "synthetic:tiny"}[args.model]
    host_only = dry and args.dry_ingest
    if host_only:      # the Inspector on the local provider with the engine stand-in: the real ingest path, no model
        from vision_inspection_system_amd.config import Qwen2VLConfig
        from vision_inspection_system_amd.tokenizer import ByteTokenizer
        model_id = "synthetic:host-only"
        hcfg = Qwen2VLConfig.qwen2_vl_7b()
        tok = ByteTokenizer(hcfg.vocab, hcfg.image_token_id, hcfg.vision_start_id, hcfg.vision_end_id, hcfg.eos_ids)
        CL.register_model(model_id, "cuda:0", CL.LoadedModel(
            HostOnlyEngine(hcfg, max(1, min(64, int(os.environ.get("VIS_MAX_BATCH", "64"))))), tok, hcfg, model_id))
    os.environ.setdefault("VIS_TINY_MAX_CTX", "4096")   # the tiny model's byte tokenizer spends one token per prompt character
    new = args.new_tokens if args.model != "tiny" else min(args.new_tokens, 24)
    local_aud = args.auditor == "mllama" and not dry
    C.set_config(C.Config(
        vlm_inspector_provider="mock" if (dry and not host_only) else "mi355x", vlm_inspector_model=model_id, vlm_inspector_max_tokens=new,
        vlm_inspector_temperature=0.0, vlm_auditor_provider="mi355x" if local_aud else "mock",
        vlm_auditor_model=("synthetic:mllama-11b" if args.model != "tiny" else "synthetic:mllama-tiny") if local_aud else "mock",
        vlm_auditor_max_tokens=new, vlm_auditor_temperature=0.0, max_image_dimension=2048))
    size = args.image_size if args.model != "tiny" else min(args.image_size, 128)
    tmp = None
    if rank == 0:
        tmp = tempfile.mkdtemp(prefix="vis_batch256_")
        for i in range(args.images):
            rng = np.random.default_rng(1234 + i)
            Image.fromarray(rng.integers(0, 256, (size, size, 3), dtype=np.uint8)).save(
                os.path.join(tmp, f"frame{i:04d}.png"), compress_level=1)
    d = agree_on(tmp or "", "batch256_dir")             # written before it is published: the files are complete
    paths = [os.path.join(d, f"frame{i:04d}.png") for i in range(args.images)]
    try:
        # warm-up = one call of the same shape (models, pools, and the decode graph of the per-rank group size - a graph is
        # captured per batch size, 0.4 s at 7B shapes); lists longer than one group per rank warm up on one group per rank
        group = int(os.environ.get("VIS_MAX_BATCH", "64"))
        run_batch_inspection(paths[:min(args.images, group * world)], "medium", "general")
        clear_encode_cache()
        CL.TIMING_LOG.clear()
        if world > 1:
            (_cpu_barrier(dist) if dry else dist.barrier())
        if not dry:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = run_batch_inspection(paths, "medium", "general")
        if not dry:
            torch.cuda.synchronize()
        mine = time.perf_counter() - t0
        if world > 1:
            (_cpu_barrier(dist) if dry else dist.barrier())
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64,
                             device="cpu" if (dry or args.share_gpu) else torch.device("cuda", local_rank))
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        dev_ms = sum(x.get("prefill_ms", 0.0) + x.get("decode_ms", 0.0) for x in CL.TIMING_LOG)
        per_rank = gather_records([{"rank": rank, "call_s": mine, "images": len(range(rank, args.images, world)),
                                    "engine_device_s": dev_ms * 1e-3, "engine_share_of_call": dev_ms * 1e-3 / max(mine, 1e-9),
                                    "groups": len(CL.TIMING_LOG)}], world)
        if rank == 0:
            done = out["session_results"]["completed_images"]
            if len(out["image_results"]) != args.images:
                raise SystemExit(f"bench.py: {len(out['image_results'])} image results, expected {args.images}")
            print(json.dumps({
                "metric": "inspected images/sec (1024x1024, Qwen2-VL-7B), run_batch_inspection over the whole seam"
                if args.model == "7b" and not dry else "images/sec (rehearsal)",
                "value": args.images / elapsed, "unit": "images/s", "n_gpus": world, "steps": 1, "warmup": 1,
                "ms_per_step": elapsed * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                "dtype": "none" if dry else "bf16", "data": "synthetic", "dry": dry,
                "config": {"workload": f"configs[3]: run_batch_inspection, {args.images} x {size}x{size} PNG files sharded "
                                       f"paths[r::{world}] over {world} rank(s), one gather of the records; Inspector "
                                       f"{('host-only stand-in on the local provider: real a3 encode + base64 / Huffman decode on the ingest pool, no model' if host_only else 'canned (no model)') if dry else model_id}, Auditor {args.auditor if not dry else 'canned'}",
                           "images": args.images, "completed": done, "new_tokens": new,
                           "ingest_threads_per_rank": int(os.environ["VIS_INGEST_THREADS"]), "host_cpus": cores,
                           "parallelism": f"dp{world} (whole images)" if not args.share_gpu else
                           f"REHEARSAL: {world} ranks sharing ONE GPU, records over gloo - not a scaling measurement"},
                "per_rank": sorted(per_rank, key=lambda r: r["rank"]),
                "note": "engine_share_of_call = device time of the rank's prompt passes + decode loops / its wall time of "
                        "the call: what is left is ingest (a3 encode / decode) the pool did not hide, host launch work "
                        "and the gather"}), flush=True)
    finally:
        if world > 1:
            (_cpu_barrier(dist) if dry else dist.barrier())
        if rank == 0 and tmp:
            import shutil
            shutil.rmtree(tmp, ignore_errors=True)
        CL.set_mock_reply(None)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--model", default="7b", choices=["7b", "tiny", "7b25"],
                    help="7b: Qwen2-VL-7B (BASELINE's headline model); 7b25: Qwen2.5-VL-7B (the reference's code default: "
                         "windowed vision tower) - NOT the headline model, the JSON says so")
    ap.add_argument("--image-size", type=int, default=1024)
    ap.add_argument("--prompt-tokens", type=int, default=1024)
    ap.add_argument("--new-tokens", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the microbench / e2e / plumbing blocks")
    ap.add_argument("--no-blocks", action="store_true",
                    help="skip the batch64 / seam64 / dual / fp8_batch4 blocks (configs[2]/[3]/[4] per-GPU slices, ~2 min)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL, the product path) or gloo (CPU rehearsal, with --dry-device cpu)")
    ap.add_argument("--dry-device", default=None, choices=["cpu"],
                    help="cpu: no model, no GPU - exercise launcher, rendezvous, record gather and timing only")
    ap.add_argument("--share-gpu", action="store_true",
                    help="REHEARSAL on a one-GPU box: all N ranks run the real model on cuda:0 and exchange their records over "
                         "gloo (RCCL cannot place two ranks on one device).  Exercises launcher, per-rank engines, sharding, "
                         "the store-voted gather and the timing protocol with real GPU work; the figure is NOT a scaling number "
                         "and the JSON says so")
    ap.add_argument("--decode-weights", default="bf16", choices=["bf16", "fp8"],
                    help="fp8: BASELINE configs[4] slice (e4m3 weights for the single-sequence decode GEMVs) - NOT the "
                         "headline precision; the JSON says so in dtype/config")
    ap.add_argument("--prefill-dtype", default="bf16", choices=["bf16", "fp8"],
                    help="fp8: BASELINE configs[4] (LLM prompt-pass projections on the fp8 MFMA) - NOT the headline precision")
    ap.add_argument("--prompt-order", default="image-first", choices=["image-first", "text-first"],
                    help="text-first: the reference's message order (text part, then image part); with --batch > 1 the "
                         "common text prefix is then computed once per batch")
    ap.add_argument("--workload", default="step", choices=["step", "batch256"],
                    help="batch256: BASELINE configs[3] as a workload - rank 0 writes --images seeded 1024x1024 PNG files, "
                         "every rank calls run_batch_inspection on the FULL list (strong scaling: rank r inspects "
                         "paths[r::W], whole seam per rank - a3 encode, JPEG decode, ingest pool, Inspector engine, parse, "
                         "consensus, gates - and ONE gather of the records), timed across the call")
    ap.add_argument("--images", type=int, default=256, help="--workload batch256: number of PNG files")
    ap.add_argument("--dry-ingest", action="store_true",
                    help="--workload batch256 --dry-device cpu: the Inspector on the LOCAL provider with an engine stand-in "
                         "(HostOnlyEngine): real a3 encode + base64 / Huffman decode on the ingest pool, no model, no GPU - "
                         "the host ceiling of a rank")
    ap.add_argument("--auditor", default="mock", choices=["mock", "mllama"],
                    help="--workload batch256: Auditor on the canned-response client (Inspector-only timing) or on the "
                         "local Llama-3.2-11B-Vision engine (dual-VLM batch inspection)")
    ap.add_argument("--batch", type=int, default=1,
                    help="images per step per GPU; > 1 uses the batched decode path (BASELINE configs[3]/[4] style "
                         "batch inspection) - NOT the headline single-image configuration")
    args = ap.parse_args()

    if args.dry_device == "cpu":
        # a dry run must not open the GPU at all (torch.cuda.is_available() would, in every rank): hide it before any
        # CUDA / HIP call - the variables are inherited by the ranks the launcher starts
        os.environ["HIP_VISIBLE_DEVICES"] = ""
        os.environ["CUDA_VISIBLE_DEVICES"] = ""
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # nobody launched ranks for us: do it here, BEFORE any GPU call (see launch_ranks)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.workload == "batch256":
        return run_batch256(args, rank, world, local_rank)
    if args.dry_device == "cpu":
        return run_dry(args, rank, world)
    if args.backend != "nccl" and not args.share_gpu:
        raise SystemExit("bench.py: --backend gloo is only for --dry-device cpu or --share-gpu; the product path is RCCL")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the hot path has no CPU fallback")
    if args.share_gpu:
        local_rank = 0                       # rehearsal: every rank on the one GPU of the box
        os.environ["VIS_DECODE_CHAIN"] = "0"  # two PROCESSES' chained launches cannot both be resident on one GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        from datetime import timedelta
        if args.share_gpu:
            dist.init_process_group("gloo", timeout=timedelta(seconds=900))
        else:
            dist.init_process_group("nccl", device_id=dev, timeout=timedelta(seconds=900))  # nccl == RCCL on ROCm

    from vision_inspection_system_amd.config import Qwen2VLConfig
    from vision_inspection_system_amd.engine import Qwen2VLEngine
    from vision_inspection_system_amd.weights import random_device_weights
    from vision_inspection_system_amd.batch import gather_records
    from vision_inspection_system_amd import hip as hip_mod

    cfg = {"7b": Qwen2VLConfig.qwen2_vl_7b, "7b25": Qwen2VLConfig.qwen2_5_vl_7b, "tiny": Qwen2VLConfig.tiny}[args.model]()
    weights = random_device_weights(cfg, dev, seed=0)
    engine = Qwen2VLEngine(cfg, weights, dev, max_ctx=4096 if args.model != "tiny" else 1024, max_batch=args.batch,
                           decode_weights=args.decode_weights, prefill_dtype=args.prefill_dtype)

    frame_np = synthetic_frame(rank, args.image_size)
    n_patches = (frame_np.shape[0] // cfg.patch) * (frame_np.shape[1] // cfg.patch)
    n_img_tok = n_patches // cfg.merge ** 2
    ids = synthetic_prompt(cfg, n_img_tok, args.prompt_tokens, order=args.prompt_order)
    S = len(ids)
    frame = torch.from_numpy(frame_np).to(dev)
    ids_dev = torch.tensor(ids, dtype=torch.int32, device=dev)
    new = args.new_tokens
    B = args.batch

    pre_ev, step_records = [], []

    def one_step_batched(timed: bool):
        s = torch.cuda.Event(enable_timing=True)
        m = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        engine.prefill_many([(ids, [frame])] * B, max_new_tokens=new, ids_dev=[ids_dev] * B)
        m.record()
        g = engine._ensure_graph(B) if not args.no_graph else None
        for _ in range(new - 1):
            if g is not None:
                g.replay()
            else:
                engine._decode_step_batched(B)
        e.record()
        toks = engine.tokens_b[:B, S - 1:S - 1 + new].cpu().tolist()
        rec = gather_records([{"image": f"synthetic_{rank}_{b}", "tokens": toks[b]} for b in range(B)], world)
        if timed:
            pre_ev.append((s, m, e))
            step_records.append(len(rec))

    def one_step(timed: bool):
        if B > 1:
            return one_step_batched(timed)
        s = torch.cuda.Event(enable_timing=True)
        m = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        engine.prefill(ids, [frame], ids_dev=ids_dev, max_new_tokens=new)
        m.record()
        engine.decode(new - 1, use_graph=not args.no_graph)
        e.record()
        try:
            toks = engine.generated(new)  # D2H of the 128 token ids (synchronises)
        except hip_mod.ChainStalled as ex:   # something else used this GPU's chained launches: same step on the four launches
            print(f"[bench] {ex}: decode chain switched off, step repeated", file=sys.stderr, flush=True)
            engine.disable_chain()
            return one_step(timed)
        rec = gather_records([{"image": f"synthetic_{rank}", "tokens": toks}], world)
        if timed:
            pre_ev.append((s, m, e))
            step_records.append(len(rec))

    for _ in range(args.warmup):
        one_step(False)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step(True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.share_gpu else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        if any(n != world * B for n in step_records):
            raise SystemExit(f"bench.py: a step gathered {step_records} records, expected {world * B} each")
        t_pre = sum(s.elapsed_time(m) for s, m, e in pre_ev) / len(pre_ev) * 1e-3      # all B prompt passes of a step
        t_dec = sum(m.elapsed_time(e) for s, m, e in pre_ev) / len(pre_ev) * 1e-3
        fp8 = args.decode_weights == "fp8"
        p8 = args.prefill_dtype == "fp8"
        if B > 1:
            k_avg, k_launches = measure_decode_gemm(engine, B)
            kernel = ("decode_proj_kernel" if engine.fused_proj else "gemm_decode_stream_kernel") + \
                ("<fp8>" if fp8 and engine.fp8_batched else "") + \
                f" (batched decode projection, weights streamed once for {B} sequences" + \
                (", split-K reduction and epilogue in the same launch)" if engine.fused_proj else ")")
        else:
            k_avg, k_launches = measure_gemv(engine)
            kernel = ("gemv_fp8w_kernel" if fp8 else "gemv_bf16_kernel") + " (decode weight streaming)"
        step_bytes = gemv_bytes_per_step(cfg) / (2 if fp8 else 1)
        if B > 1 and fp8 and engine.fp8_batched:      # the o projection stays bf16 in the batched fp8 step
            step_bytes += cfg.layers * cfg.heads * cfg.head_dim * cfg.hidden
        kernel_bytes = step_bytes
        if B == 1 and engine.chain_sync is not None:  # qkv / o weights stream inside the chained layer-head launch
            D_ = cfg.head_dim
            kernel_bytes -= 2.0 * cfg.layers * (cfg.hidden * (cfg.heads + 2 * cfg.kv_heads) * D_ + cfg.heads * D_ * cfg.hidden)
            kernel_bytes -= 2.0 * cfg.hidden * cfg.vocab
            kernel = "gemv_bf16_kernel<1, false> (decode weight streaming: the gate/up and down projections, 81 % of a token's " \
                     "weight bytes; qkv / o run inside decode_chain_kernel, the lm_head as gemv_bf16_kernel<1, true>)"
        bytes_per_launch = kernel_bytes / k_launches
        achieved = bytes_per_launch / k_avg / 1e9
        flops = prefill_flops(cfg, n_patches, S) * B          # B full prompt passes' worth of arithmetic per step
        out = {
            "metric": {"7b": "inspected images/sec (1024x1024, Qwen2-VL-7B)",
                       "7b25": "inspected images/sec (1024x1024, Qwen2.5-VL-7B - NOT the headline model)"}.get(args.model, "images/sec (tiny)"),
            "value": world * args.steps * B / elapsed, "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if not (fp8 or p8) else
            f"{'fp8-e4m3 (LLM projections)' if p8 else 'bf16'} prefill / {'fp8-e4m3' if fp8 else 'bf16'} decode weights",
            "data": "synthetic",
            "config": {"workload": (f"configs[1]: Qwen2-VL-7B Inspector bf16, single {args.image_size}x"
                                    f"{args.image_size} image per step per GPU, greedy decode {new} tok")
                       if B == 1 and not (fp8 or p8) else
                       (f"configs[4] (NOT the headline precision): Qwen2-VL-7B, {'fp8 MFMA' if p8 else 'bf16'} LLM prefill, "
                        f"{'fp8-e4m3' if fp8 else 'bf16'} weights in the decode GEMVs, single {args.image_size}x"
                        f"{args.image_size} image, greedy decode {new} tok")
                       if B == 1 else
                       (f"batch inspection: {B} x {args.image_size}x{args.image_size} images per step per GPU, "
                        f"per-image prefill + batched decode {new} tok (NOT the headline configuration)"
                        + ("; text part first as in the reference: the shared text prefix is computed once per batch"
                           if args.prompt_order == "text-first" else "")),
                       "batch": B,
                       "image_px": args.image_size, "resized_px": list(frame_np.shape[:2]),
                       "image_tokens": n_img_tok, "prompt_tokens": S, "new_tokens": new,
                       "weights": "seeded random bf16 at exact 7B shapes",
                       "parallelism": f"dp{world} (whole images)" if not args.share_gpu else
                       f"REHEARSAL: {world} ranks sharing ONE GPU, records over gloo - not a scaling measurement"},
            "roofline": {"bound": "hbm", "kernel": kernel,
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None if (fp8 or B > 1) else measured_traffic("gemv", True)[0],
                         # not measured in this run: read from the tracked rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
                         "traffic_source": None if (fp8 or B > 1) else measured_traffic("gemv", True)[1],
                         "bytes_per_launch": bytes_per_launch,
                         "avg_launch_us": k_avg * 1e6, "launches_per_token": k_launches,
                         # the same bound over the WHOLE decode token (every weight byte of a step / the step's time): the
                         # figure that stays comparable when the set of kernels behind `kernel` changes between rounds
                         "whole_token_frac": step_bytes * (new - 1) / t_dec / 1e9 / HBM_PEAK_GBS},
            "prefill_mfma": {"flops": flops, "ms": t_pre * 1e3, "images": B, "achieved": flops / t_pre / 1e12,
                             "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s", "frac": flops / t_pre / 1e12 / MFMA_BF16_PEAK_TF,
                             "note": ("FLOPs of B full prompt passes / time of the B prompt passes of a step; with a "
                                      "shared text prefix less arithmetic is actually executed, so this is an "
                                      "effective rate") if B > 1 else "one prompt pass"},
            "decode": {"ms": t_dec * 1e3, "ms_per_token": t_dec / (new - 1) * 1e3,
                       "chained_layer_head": bool(B == 1 and engine.chain_sync is not None),
                       "weight_GBps": step_bytes * (new - 1) / t_dec / 1e9,
                       "sequences_per_step": B},
        }
        extras = not args.no_extras and args.model == "7b" and world == 1
        if extras:
            mb = microbench(dev)
            best = max(v for k, v in mb.items() if k.startswith("gemm_") and v)
            mb["prefill_frac_of_measured_gemm"] = out["prefill_mfma"]["achieved"] / best
            mb["roofline_frac_of_measured_copy"] = achieved / mb["stream_copy_GBps"]
            if mb.get("stream_read_GBps"):
                mb["roofline_frac_of_measured_read"] = achieved / mb["stream_read_GBps"]
            out["microbench"] = mb
            if B == 1 and not (fp8 or p8):
                out["e2e"] = e2e_request(engine, cfg, args.image_size, args.prompt_tokens, new)
                out["e2e"]["kernel_only_ms"] = elapsed / args.steps * 1e3
            out["plumbing"] = plumbing_baseline()
            if B == 1 and not (fp8 or p8) and not args.no_blocks:
                # configs[2] / [3] / [4] per-GPU slices, each with its own events and roofline (tools/bench_blocks.py); the
                # headline value / config above stay configs[1]
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import bench_blocks
                out.update(bench_blocks.run_all(cfg, weights, dev, frame, n_patches, n_img_tok, new, args.prompt_tokens,
                                                log=lambda m: print(f"[bench] {m}", file=sys.stderr, flush=True)))
        if not args.no_cpu_baseline and args.model == "7b" and world == 1:     # rank 0 at N = 1 only (the other ranks would idle)
            out["cpu_baseline"] = cpu_baseline(cfg, n_patches, S, new)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
