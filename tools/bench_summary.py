"""One-screen summary of a bench.py JSON line: python tools/bench_summary.py FILE"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"headline {d['value']:.4f} {d['unit']}  {d['ms_per_step']:.1f} ms/step  roofline.frac {d['roofline']['frac']:.4f}")
if "prefill_mfma" in d:
    print(f"prompt pass {d['prefill_mfma']['ms']:.2f} ms  frac {d['prefill_mfma']['frac']:.4f}   decode {d['decode']['ms_per_token']:.4f} ms/token")
for k in ("batch64", "seam64", "fp8_batch4"):
    b = d.get(k)
    if b:
        print(k, " ".join(f"{kk}={b[kk]:.3f}" for kk in ("images_per_s", "prompt_pass_ms_per_image", "decode_ms_per_step") if kk in b),
              "roofline.frac=%.3f" % b["roofline"]["frac"] if "roofline" in b else "")
if "dual" in d:
    print("dual single %.3f images/s, batch32 %.3f images/s" % (d["dual"]["single"]["images_per_s"], d["dual"]["batch32"]["images_per_s"]))
if "e2e" in d:
    print("e2e warm %.1f ms, cold %.1f ms" % (d["e2e"]["ms"], d["e2e"]["cold_prefix_ms"]))
