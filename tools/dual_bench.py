"""BASELINE configs[2]: dual-VLM inspection of ONE 1024x1024 image on one GPU - Inspector (Qwen2-VL-7B) and Auditor
(Llama-3.2-11B-Vision) replicas resident together, each prefill + 128 greedy tokens, then the reference's
post-processing (parse -> validate -> consensus -> safety gates) on the two replies.  Seeded random weights at the
exact shapes (no checkpoints offline): timing is valid, the replies are noise and take the documented failure path.
Prints one JSON line."""
import argparse, json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vision_inspection_system_amd.config import Qwen2VLConfig
from vision_inspection_system_amd.engine import Qwen2VLEngine
from vision_inspection_system_amd.weights import random_device_weights
from vision_inspection_system_amd import mllama_weights as MW
from vision_inspection_system_amd.mllama_engine import MllamaEngine
from vision_inspection_system_amd.image_processing import smart_resize
from vision_inspection_system_amd.consensus import analyze_consensus
from vision_inspection_system_amd.gates import evaluate_safety
from vision_inspection_system_amd.schemas import InspectionContext, VLMAnalysisResult
from vision_inspection_system_amd import hip

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--new-tokens", type=int, default=128)
ap.add_argument("--prompt-tokens", type=int, default=700)
ap.add_argument("--batch", type=int, default=1, help="> 1: batch inspection - that many images per step through BOTH models, "
                "each model: per-image prompt pass + one shared decode loop (text part first: shared prefix for the Inspector)")
a = ap.parse_args()
dev = torch.device("cuda:0")
qc = Qwen2VLConfig.qwen2_vl_7b()
B = a.batch
insp = Qwen2VLEngine(qc, random_device_weights(qc, dev, 0), dev, max_ctx=4096, max_batch=B)
mc = MW.MllamaConfig.mllama_11b()
aud = MllamaEngine(mc, MW.random_device_weights(mc, dev, 1), dev, max_ctx=2048, max_batch=B)
rng = np.random.default_rng(0)
raw = torch.from_numpy(rng.integers(0, 256, (1024, 1024, 3), dtype=np.uint8)).to(dev)
th, tw = smart_resize(1024, 1024)
n_img = (th // 14) * (tw // 14) // 4
q_text = rng.integers(0, 1000, a.prompt_tokens).tolist()
q_ids = [qc.vision_start_id] + [qc.image_token_id] * n_img + [qc.vision_end_id] + q_text
if B > 1:   # the reference's part order: inspection prompt first, then the image (vlm_inspector.py:462-470)
    q_ids = q_text + [qc.vision_start_id] + [qc.image_token_id] * n_img + [qc.vision_end_id] + [5, 6]
m_ids = [1] + rng.integers(1000, mc.vocab - 8, a.prompt_tokens).tolist() + [mc.image_token_id, 5, 6]
ctx = InspectionContext(image_id="bench", criticality="medium")
times = []
for it in range(a.steps + 1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    frame = hip.resize_rgb(raw, th, tw)                       # Inspector frame (bicubic, GPU)
    if B > 1:
        it_tok = insp.generate_batch([(q_ids, [frame])] * B, max_new_tokens=a.new_tokens, ignore_eos=True)
        au_tok = aud.generate_batch([(m_ids, raw)] * B, max_new_tokens=a.new_tokens, stop_on_eos=False)
    else:
        insp.prefill(q_ids, [frame], max_new_tokens=a.new_tokens)
        insp.decode(a.new_tokens - 1)
        aud.prefill(m_ids, raw)                                   # Auditor: tiles (bilinear, GPU)
        aud.decode(a.new_tokens - 1)
        it_tok, au_tok = insp.generated(a.new_tokens), aud.generated(a.new_tokens)   # D2H sync
    t1 = time.perf_counter()
    r1 = VLMAnalysisResult(object_identified="unknown", overall_condition="uncertain", defects=[], overall_confidence="low",
                           analysis_failed=True, failure_reason="Failed to parse JSON")     # noise replies -> failure path
    for _ in range(B):
        cons = analyze_consensus(r1, r1)
        verdict = evaluate_safety(cons, ctx)
    t2 = time.perf_counter()
    if it:
        times.append((t1 - t0, t2 - t1))
gpu = float(np.mean([t[0] for t in times])); host = float(np.mean([t[1] for t in times]))
print(json.dumps({"workload": "configs[2]: dual-VLM Qwen2-VL-7B + Llama-3.2-11B-Vision, one 1024x1024 image, 128+128 greedy tokens, consensus + gates",
                  "batch": B, "ms_per_image": (gpu + host) * 1e3 / B, "images_per_s": B / (gpu + host), "gpu_ms": gpu * 1e3,
                  "postprocess_ms": host * 1e3, "verdict": verdict.verdict,
                  "resident_GB": torch.cuda.memory_allocated() / 1e9}))
