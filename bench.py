#!/usr/bin/env python3
"""Headline benchmark: inspected images/sec, Qwen2-VL-7B, synthetic 1024x1024 frames (BASELINE.json).

A "step" is one pass of the hot path over one image: K1 patchify -> ViT -> merger -> LLM prefill over
S = 2249 tokens (1225 image + 1024 text) -> exactly 128 greedy tokens (EOS ignored), the configuration
BASELINE.json's metric is quoted on (configs[1]).  Inputs (resized u8 frame, token ids) are resident in
HBM before the timed region.  Weights are seeded random bf16 at the exact 7B shapes (no checkpoint exists
offline), so timing is valid and the generated text is noise.

    python bench.py --gpus N --steps K --warmup W

N > 1 is launched by torch.distributed.run (one rank per GPU, RCCL): whole images are sharded across
ranks (weak scaling: every rank inspects its own image stream), each step ends with one RCCL all_gather
of the per-image result records, and the reported time is the max over ranks.

The single JSON line also carries
  roofline     - the dominant kernel (decode weight-streaming GEMV, HBM-bound): algorithmic bytes per
                 launch / average launch duration, measured live with HIP events;
  prefill_mfma - prefill FLOPs / prefill time against the dense bf16 MFMA peak;
  cpu_baseline - the oracle (CPU port of the same arithmetic) timed on this host's cores on a bounded
                 sample of the same workload.
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
    vit_gemm = 2.0 * n_patches * (cfg.patch_dim * E + cfg.v_depth * (3 * E * E + E * E + 2 * E * cfg.v_mlp))
    vit_gemm += 2.0 * (n_patches // cfg.merge ** 2) * (M * M + M * cfg.hidden)
    vit_attn = cfg.v_depth * 4.0 * n_patches * n_patches * E
    D = cfg.head_dim
    per_layer = cfg.hidden * (cfg.heads + 2 * cfg.kv_heads) * D + cfg.heads * D * cfg.hidden \
        + 3 * cfg.hidden * cfg.intermediate
    llm_gemm = 2.0 * S * cfg.layers * per_layer + 2.0 * cfg.hidden * cfg.vocab
    llm_attn = cfg.layers * 2.0 * S * S * cfg.heads * D  # causal: half of 4*S^2*H*D
    return vit_gemm + vit_attn + llm_gemm + llm_attn


def gemv_bytes_per_step(cfg) -> float:
    D = cfg.head_dim
    per_layer = cfg.hidden * (cfg.heads + 2 * cfg.kv_heads) * D + cfg.heads * D * cfg.hidden \
        + 3 * cfg.hidden * cfg.intermediate
    return 2.0 * (cfg.layers * per_layer + cfg.hidden * cfg.vocab)


def measure_gemv(engine, reps: int = 5):
    """Average duration of one gemv_bf16_kernel launch, measured live with HIP events on the launch stream.

    The 113 GEMV launches of one decode step (same weights, same buffers, same arguments as the real step)
    are captured alone into a hipGraph and the replay is bracketed by two events, so the figure is
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
        for lw in w.llm:
            hip.gemv(x[0], lw.qkv_w, engine.d_qkv, bias=lw.qkv_b, norm_w=lw.ln1_w, eps=cfg.rms_eps)
            hip.gemv(engine.d_attn, lw.o_w, x2[0], residual=x[0])
            hip.gemv(x2[0], lw.gateup_w, engine.d_act, norm_w=lw.ln2_w, act=hip.ACT_SWIGLU, eps=cfg.rms_eps)
            hip.gemv(engine.d_act, lw.down_w, x[0], residual=x2[0])
            n[0] += 4
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


def measured_traffic():
    """HBM bytes per gemv launch from the committed rocprofv3 PMC passes (profiles/*_gemv_traffic.json), or None."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_gemv_traffic.json")))
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            return float(json.load(f)["hbm_bytes_per_launch"])
    except Exception:
        return None


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
        pv = torch.randn(n_patches, cfg.patch_dim)
        t0 = time.perf_counter()
        R.vision_forward(rc, sd, pv, [(1, side, side)])
        t_vit = time.perf_counter() - t0           # patch embed + 1 block + merger
        x = torch.randn(S, H)
        cos, sin = R.mrope_cos_sin(rc, torch.arange(S).view(1, -1).expand(3, -1))
        cache = R.KVCache(1)
        t0 = time.perf_counter()
        R.text_forward(rc, sd, x, cos, sin, cache)
        t_llm = time.perf_counter() - t0           # 1 prefill layer
        c1, s1 = R.mrope_cos_sin(rc, torch.full((3, 1), S))
        t0 = time.perf_counter()
        n_dec = 4
        for _ in range(n_dec):
            h = R.text_forward(rc, sd, torch.randn(1, H), c1, s1, cache)
        t_dec = (time.perf_counter() - t0) / n_dec  # 1 decode layer
        t0 = time.perf_counter()
        for _ in range(2):
            _ = h[-1] @ lm_head.t()
        t_lm = (time.perf_counter() - t0) / 2
    t_img = cfg.v_depth * t_vit + cfg.layers * t_llm + new_tokens * (cfg.layers * t_dec + t_lm)
    return {"value": 1.0 / t_img, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": (f"oracle fp32 at 7B shapes: 1 ViT block N={n_patches} ({t_vit:.2f}s), 1 LLM prefill layer "
                       f"S={S} ({t_llm:.2f}s), {n_dec} decode-layer steps ({t_dec * 1e3:.0f}ms each), lm_head "
                       f"({t_lm * 1e3:.0f}ms); extrapolated x{cfg.v_depth}/x{cfg.layers}/x{new_tokens} to one image "
                       f"= {t_img:.1f}s. The reference has no CPU arithmetic path (remote API).")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--model", default="7b", choices=["7b", "tiny"])
    ap.add_argument("--image-size", type=int, default=1024)
    ap.add_argument("--prompt-tokens", type=int, default=1024)
    ap.add_argument("--new-tokens", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--decode-weights", default="bf16", choices=["bf16", "fp8"],
                    help="fp8: BASELINE configs[4] slice (e4m3 weights for the single-sequence decode GEMVs) - NOT the "
                         "headline precision; the JSON says so in dtype/config")
    ap.add_argument("--prefill-dtype", default="bf16", choices=["bf16", "fp8"],
                    help="fp8: BASELINE configs[4] (LLM prompt-pass projections on the fp8 MFMA) - NOT the headline precision")
    ap.add_argument("--prompt-order", default="image-first", choices=["image-first", "text-first"],
                    help="text-first: the reference's message order (text part, then image part); with --batch > 1 the "
                         "common text prefix is then computed once per batch")
    ap.add_argument("--batch", type=int, default=1,
                    help="images per step per GPU; > 1 uses the batched decode path (BASELINE configs[3]/[4] style "
                         "batch inspection) - NOT the headline single-image configuration")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm

    from vision_inspection_system_amd.config import Qwen2VLConfig
    from vision_inspection_system_amd.engine import Qwen2VLEngine
    from vision_inspection_system_amd.weights import random_device_weights
    from vision_inspection_system_amd.batch import gather_records

    cfg = Qwen2VLConfig.qwen2_vl_7b() if args.model == "7b" else Qwen2VLConfig.tiny()
    weights = random_device_weights(cfg, dev, seed=0)
    engine = Qwen2VLEngine(cfg, weights, dev, max_ctx=4096 if args.model == "7b" else 1024, max_batch=args.batch,
                           decode_weights=args.decode_weights, prefill_dtype=args.prefill_dtype)

    frame_np = synthetic_frame(rank, args.image_size)
    n_patches = (frame_np.shape[0] // cfg.patch) * (frame_np.shape[1] // cfg.patch)
    n_img_tok = n_patches // cfg.merge ** 2
    ids = synthetic_prompt(cfg, n_img_tok, args.prompt_tokens, order=args.prompt_order)
    S = len(ids)
    frame = torch.from_numpy(frame_np).to(dev)
    ids_dev = torch.tensor(ids, dtype=torch.int32, device=dev)
    new = args.new_tokens

    pre_ev, step_records = [], []

    def one_step_batched(timed: bool):
        s = torch.cuda.Event(enable_timing=True)
        m = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        B = args.batch
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
        if args.batch > 1:
            return one_step_batched(timed)
        s = torch.cuda.Event(enable_timing=True)
        m = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        engine.prefill(ids, [frame], ids_dev=ids_dev, max_new_tokens=new)
        m.record()
        engine.decode(new - 1, use_graph=not args.no_graph)
        e.record()
        toks = engine.generated(new)  # D2H of the 128 token ids (synchronises)
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
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        t_pre = sum(s.elapsed_time(m) for s, m, e in pre_ev) / len(pre_ev) * 1e-3
        t_dec = sum(m.elapsed_time(e) for s, m, e in pre_ev) / len(pre_ev) * 1e-3
        gemv_avg, gemv_launches = measure_gemv(engine)
        fp8 = args.decode_weights == "fp8"
        p8 = args.prefill_dtype == "fp8"
        step_bytes = gemv_bytes_per_step(cfg) / (2 if fp8 else 1)
        bytes_per_launch = step_bytes / gemv_launches
        achieved = bytes_per_launch / gemv_avg / 1e9
        flops = prefill_flops(cfg, n_patches, S)
        out = {
            "metric": "inspected images/sec (1024x1024, Qwen2-VL-7B)" if args.model == "7b" else "images/sec (tiny)",
            "value": world * args.steps * args.batch / elapsed, "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if not (fp8 or p8) else
            f"{'fp8-e4m3 (LLM projections)' if p8 else 'bf16'} prefill / {'fp8-e4m3' if fp8 else 'bf16'} decode weights",
            "data": "synthetic",
            "config": {"workload": (f"configs[1]: Qwen2-VL-7B Inspector bf16, single {args.image_size}x"
                                    f"{args.image_size} image per step per GPU, greedy decode {new} tok")
                       if args.batch == 1 and not (fp8 or p8) else
                       (f"configs[4] (NOT the headline precision): Qwen2-VL-7B, {'fp8 MFMA' if p8 else 'bf16'} LLM prefill, "
                        f"{'fp8-e4m3' if fp8 else 'bf16'} weights in the decode GEMVs, single {args.image_size}x"
                        f"{args.image_size} image, greedy decode {new} tok")
                       if args.batch == 1 else
                       (f"batch inspection: {args.batch} x {args.image_size}x{args.image_size} images per step per GPU, "
                        f"per-image prefill + batched decode {new} tok (NOT the headline configuration)"
                        + ("; text part first as in the reference: the shared text prefix is computed once per batch"
                           if args.prompt_order == "text-first" else "")),
                       "batch": args.batch,
                       "image_px": args.image_size, "resized_px": list(frame_np.shape[:2]),
                       "image_tokens": n_img_tok, "prompt_tokens": S, "new_tokens": new,
                       "weights": "seeded random bf16 at exact 7B shapes", "parallelism": f"dp{world} (whole images)"},
            "roofline": {"bound": "hbm", "kernel": ("gemv_fp8w_kernel" if fp8 else "gemv_bf16_kernel") +
                         " (decode weight streaming)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None if fp8 else measured_traffic(), "bytes_per_launch": bytes_per_launch, "avg_launch_us": gemv_avg * 1e6,
                         "launches_per_token": gemv_launches},
            "prefill_mfma": {"flops": flops, "ms": t_pre * 1e3, "achieved": flops / t_pre / 1e12,
                             "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s", "frac": flops / t_pre / 1e12 / MFMA_BF16_PEAK_TF},
            "decode": {"ms": t_dec * 1e3, "ms_per_token": t_dec / (new - 1) * 1e3,
                       "weight_GBps": step_bytes * (new - 1) / t_dec / 1e9,
                       "sequences_per_step": args.batch},
        }
        if not args.no_cpu_baseline and args.model == "7b" and world == 1:     # rank 0 at N = 1 only (the other ranks would idle)
            out["cpu_baseline"] = cpu_baseline(cfg, n_patches, S, new)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
