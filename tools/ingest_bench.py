"""Host ingest vs kernel-only rate: run_batch_inspection on real PNG files through the whole seam -
PIL open / thumbnail / JPEG q85 / base64 per agent (a3), data-URI decode, GPU resize, tokenise, per-image prompt pass,
shared decode loop, detokenise, parse, consensus, gates, aggregation - with the Inspector on synthetic:7b and the Auditor
on synthetic:mllama-11b (or the canned mock), 128 new tokens per model (EOS ignored: random weights), one rank.
Prints one JSON line per ingest-thread setting.

  python tools/ingest_bench.py --images 64 --auditor mllama|mock --threads 1,4,16
"""
import argparse, json, os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--images", type=int, default=64)
ap.add_argument("--size", type=int, default=1024)
ap.add_argument("--auditor", default="mllama", choices=["mllama", "mock"])
ap.add_argument("--threads", default="4")
ap.add_argument("--new-tokens", type=int, default=128)
ap.add_argument("--direct", action="store_true", help="VIS_DIRECT_FRAMES=1: no JPEG round trip between agent and engine")
ap.add_argument("--profile", action="store_true", help="cProfile the measured call (main thread) and print the top entries")
ap.add_argument("--switch-interval", type=float, default=0.0, help="sys.setswitchinterval (s) for the measured runs; 0 = leave")
ap.add_argument("--repeat", type=int, default=1)
a = ap.parse_args()
os.environ["VIS_IGNORE_EOS"] = "1"
if a.direct:
    os.environ["VIS_DIRECT_FRAMES"] = "1"
# random weights generate noise; a fixed parseable reply (substituted after the full generation) keeps the agents on their
# success path instead of the failure + retry-with-back-off path
os.environ["VIS_SYNTHETIC_REPLY"] = ('{"object_identified": "part", "overall_condition": "good", "defects": [], '
                                     '"overall_confidence": "high", "analysis_reasoning": "no visible damage"}')
os.environ.setdefault("VIS_MAX_BATCH", "64")
os.environ.setdefault("VIS_MAX_CTX", "4096")
from PIL import Image
from vision_inspection_system_amd import client as CL, config as C, ingest
from vision_inspection_system_amd.batch import run_batch_inspection
CL.set_mock_reply('{"object_identified": "part", "overall_condition": "good", "defects": [], "overall_confidence": "high", "analysis_reasoning": "ok"}')
cfg = C.Config(vlm_inspector_provider="mi355x", vlm_inspector_model="synthetic:7b", vlm_inspector_max_tokens=a.new_tokens,
               vlm_inspector_temperature=0.0,
               vlm_auditor_provider="mi355x" if a.auditor == "mllama" else "mock",
               vlm_auditor_model="synthetic:mllama-11b" if a.auditor == "mllama" else "mock", vlm_auditor_max_tokens=a.new_tokens,
               vlm_auditor_temperature=0.0)
C.set_config(cfg)
with tempfile.TemporaryDirectory() as d:
    paths = []
    for i in range(a.images):
        rng = np.random.default_rng(1234 + i)
        p = os.path.join(d, f"frame{i:03d}.png")
        Image.fromarray(rng.integers(0, 256, (a.size, a.size, 3), dtype=np.uint8)).save(p)
        paths.append(p)
    run_batch_inspection(paths[:4], "medium", "general")                    # loads the models, warms the graphs
    if a.switch_interval > 0:
        sys.setswitchinterval(a.switch_interval)
    for n in [int(x) for x in a.threads.split(",")] * a.repeat:
        os.environ["VIS_INGEST_THREADS"] = str(n)
        ingest.shutdown()
        from vision_inspection_system_amd.image_processing import clear_encode_cache
        clear_encode_cache()            # every measured run encodes its own images (the two agents still share one encode)
        if a.profile:
            import cProfile, pstats
            pr = cProfile.Profile()
            pr.enable()
        t0 = time.perf_counter()
        out = run_batch_inspection(paths, "medium", "general")
        t = time.perf_counter() - t0
        if a.profile:
            pr.disable()
            pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
        done = out["session_results"]["completed_images"]
        print(json.dumps({"workload": f"run_batch_inspection, {a.images} PNG files {a.size}x{a.size}, Inspector synthetic:7b"
                                      f" + Auditor {a.auditor}, {a.new_tokens} tokens per model, 1 rank",
                          "ingest_threads": n, "switch_interval": sys.getswitchinterval(), "direct_frames": bool(a.direct), "images_per_s": a.images / t, "seconds": t, "completed": done}), flush=True)
