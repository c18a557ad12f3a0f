"""EXPERIMENT: Inspector (Qwen2-VL-7B) and Auditor (Llama-3.2-11B-Vision) batches of 32 on ONE GPU, one after the other vs
at the same time (two host threads, one CUDA stream each).  python tools/probes/dual_concurrent.py [B]"""
import os, sys, threading, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vision_inspection_system_amd import hip, mllama_weights as MW
from vision_inspection_system_amd.config import Qwen2VLConfig
from vision_inspection_system_amd.engine import Qwen2VLEngine
from vision_inspection_system_amd.image_processing import smart_resize
from vision_inspection_system_amd.mllama_engine import MllamaEngine
from vision_inspection_system_amd.weights import random_device_weights
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
new = 128
dev = torch.device("cuda:0")
qc = Qwen2VLConfig.qwen2_vl_7b()
insp = Qwen2VLEngine(qc, random_device_weights(qc, dev, 0), dev, max_ctx=4096, max_batch=B)
mc = MW.MllamaConfig.mllama_11b()
aud = MllamaEngine(mc, MW.random_device_weights(mc, dev, 1), dev, max_ctx=2048, max_batch=B)
rng = np.random.default_rng(0)
raw = torch.from_numpy(rng.integers(0, 256, (1024, 1024, 3), dtype=np.uint8)).to(dev)
th, tw = smart_resize(1024, 1024)
n_img = (th // qc.patch) * (tw // qc.patch) // qc.merge ** 2
q_text = rng.integers(0, 1000, 700).tolist()
q_ids = q_text + [qc.vision_start_id] + [qc.image_token_id] * n_img + [qc.vision_end_id] + [5, 6]
m_ids = [1] + rng.integers(1000, mc.vocab - 8, 700).tolist() + [mc.image_token_id, 5, 6]
frame = hip.resize_rgb(raw, th, tw)
res = {}
def run_i():
    res["i"] = insp.generate_batch([(q_ids, [frame])] * B, max_new_tokens=new, ignore_eos=True)
def run_a():
    res["a"] = aud.generate_batch([(m_ids, raw)] * B, max_new_tokens=new, stop_on_eos=False)
PRIO = os.environ.get("DUAL_PRIO") == "1"      # the Auditor's thread on a high-priority stream (another hardware queue pool)
def on_stream(fn):
    def body():
        st = torch.cuda.Stream(device=dev, priority=-1 if (PRIO and fn is run_a) else 0)
        st.wait_stream(torch.cuda.default_stream(dev))
        with torch.cuda.stream(st):
            fn()
        st.synchronize()
    return body
run_i(); run_a(); torch.cuda.synchronize()          # warm: graphs captured single-threaded
ref_i, ref_a = res["i"], res["a"]
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run_i(); run_a(); torch.cuda.synchronize()
    t_seq = time.perf_counter() - t0
    ok_seq = res["i"] == ref_i and res["a"] == ref_a
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ts = [threading.Thread(target=on_stream(run_i)), threading.Thread(target=on_stream(run_a))]
    for t in ts: t.start()
    for t in ts: t.join()
    torch.cuda.synchronize()
    t_par = time.perf_counter() - t0
    ok_par = res["i"] == ref_i and res["a"] == ref_a
    print(f"B={B}: sequential {t_seq*1e3:.0f} ms ({B/t_seq:.2f} images/s, same tokens {ok_seq})   concurrent {t_par*1e3:.0f} ms ({B/t_par:.2f} images/s, same tokens {ok_par})", flush=True)
