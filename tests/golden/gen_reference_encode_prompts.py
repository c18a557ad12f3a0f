"""DEV-ONLY generator of the a3 / a4 golden vectors (G1 encode step, G4 formatted prompts).

Imports the reference's own ``VLMInspectorAgent`` / ``VLMAuditorAgent`` (src/agents/vlm_inspector.py:46-88,
src/agents/vlm_auditor.py:85-108) and ``utils/prompts.py`` exactly as gen_reference_postprocess.py does (unmodified
reference files, third-party packages bypassed), runs the reference's ``_encode_image_optimized`` on seeded images
(recipes in tests/helpers.py, plus the 1x1 JPEG the reference's own tests use, tests/conftest.py:19-59) and
``INSPECTOR_PROMPT.format`` / ``AUDITOR_PROMPT.format`` on three contexts, and records ONLY digests: SHA-256 and
length of every data URI / formatted prompt, or the exception type.  No reference text is stored.

Usage:  python tests/golden/gen_reference_encode_prompts.py     (writes reference_encode_prompts.json)
"""
import base64
import hashlib
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

from gen_reference_postprocess import load_reference  # noqa: E402
from helpers import ENCODE_RECIPES, LARGE_RECIPES, make_recipe_image  # noqa: E402

PROMPT_CONTEXTS = [
    {"criticality": "low", "domain": None, "user_notes": None},
    {"criticality": "medium", "domain": "mechanical_fasteners", "user_notes": "check the thread root"},
    {"criticality": "high", "domain": "aerospace", "user_notes": "Operator saw {braces} and 100% glare — unicode"},
]


def digest(s: str) -> dict:
    b = s.encode("utf-8")
    return {"sha256": hashlib.sha256(b).hexdigest(), "length": len(b)}


def run_encode(fn, *a, **k) -> dict:
    try:
        return digest(fn(*a, **k))
    except Exception as e:
        return {"error": type(e).__name__}


def reference_test_jpeg() -> bytes:
    """The 1x1 JPEG fixture of the reference's own test suite (tests/conftest.py:19-59): the byte list is evaluated
    from that file at generation time and travels as data (base64) inside the JSON."""
    import ast
    import re
    src = open("/root/reference/tests/conftest.py").read()
    m = re.search(r"jpeg_bytes = bytes\((\[.*?\])\)", src, re.S)
    return bytes(ast.literal_eval(m.group(1)))


def main():
    from pathlib import Path
    insp, aud = load_reference()[:2]
    from utils.prompts import AUDITOR_PROMPT, INSPECTOR_PROMPT
    out = {"encode": [], "prompts": [], "pillow": __import__("PIL").__version__}
    with tempfile.TemporaryDirectory() as d:
        cases = []
        for r in ENCODE_RECIPES:
            p = os.path.join(d, r["name"] + (".jpg" if r["format"] == "JPEG" else ".png"))
            make_recipe_image(r, p)
            cases.append((r["name"], p, None))
        jb = reference_test_jpeg()
        p = os.path.join(d, "reference_1x1.jpg")
        open(p, "wb").write(jb)
        cases.append(("reference_1x1_jpeg", p, base64.b64encode(jb).decode()))
        for name, path, data in cases:
            rec = {"name": name,
                   "inspector_default": run_encode(insp._encode_image_optimized, Path(path)),        # max 2048
                   "inspector_max64": run_encode(insp._encode_image_optimized, Path(path), 64),
                   "auditor_default": run_encode(aud._encode_image_optimized, Path(path)),            # max 1024
                   "auditor_max300": run_encode(aud._encode_image_optimized, Path(path), 300)}
            if data is not None:
                rec["file_base64"] = data
            out["encode"].append(rec)
        # the quality-60 retry (> 5 MB at q85, vlm_inspector.py:70-74) and the "> 10 MB even at q60" refusal (:77-80)
        # need frames larger than the default 2048 cap, so max_size is passed explicitly
        out["encode_large"] = []
        for big in LARGE_RECIPES:
            p = os.path.join(d, big["name"] + ".png")
            make_recipe_image(big, p)
            out["encode_large"].append({"recipe": big, "max_size": 6000,
                                        "inspector": run_encode(insp._encode_image_optimized, Path(p), 6000),
                                        "auditor": run_encode(aud._encode_image_optimized, Path(p), 6000)})
            os.remove(p)
    for ctx in PROMPT_CONTEXTS:
        # the calls of vlm_inspector.py:452-456 and vlm_auditor.py:188-191
        ins = INSPECTOR_PROMPT.format(criticality=ctx["criticality"], domain=ctx["domain"] or "general",
                                      user_notes=ctx["user_notes"] or "None provided")
        au = AUDITOR_PROMPT.format(criticality=ctx["criticality"], domain=ctx["domain"] or "general")
        out["prompts"].append({"context": ctx, "inspector": digest(ins), "auditor": digest(au)})
    out["prompt_templates"] = {"inspector": digest(INSPECTOR_PROMPT), "auditor": digest(AUDITOR_PROMPT)}
    with open(os.path.join(HERE, "reference_encode_prompts.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(json.dumps({k: (len(v) if isinstance(v, list) else v) for k, v in out.items()}, indent=1)[:1500])


if __name__ == "__main__":
    main()
