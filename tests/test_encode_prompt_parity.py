"""Rows a3 / a4 against vectors recorded from the reference's OWN functions (tests/golden/gen_reference_encode_prompts.py
ran VLMInspectorAgent._encode_image_optimized, vlm_inspector.py:46-88, VLMAuditorAgent._encode_image_optimized,
vlm_auditor.py:85-108, and INSPECTOR_PROMPT / AUDITOR_PROMPT .format, utils/prompts.py:18-174, in the dev container):
the product's request-side encode must produce the byte-identical data URI (SHA-256 + length) or the same exception
type, and - when the host application's utils.prompts is importable - the identical formatted prompt."""
import base64
import hashlib
import importlib
import json
import os
import sys
from pathlib import Path

import PIL
import pytest

from helpers import ENCODE_RECIPES, GOLDEN, LARGE_RECIPES, make_recipe_image

VEC = json.load(open(os.path.join(GOLDEN, "reference_encode_prompts.json")))
pytestmark = pytest.mark.skipif(PIL.__version__ != VEC["pillow"],
                                reason="JPEG bytes are pinned to the Pillow build that recorded them")


def _digest(fn, *a, **k):
    try:
        b = fn(*a, **k).encode("utf-8")
        return {"sha256": hashlib.sha256(b).hexdigest(), "length": len(b)}
    except Exception as e:
        return {"error": type(e).__name__}


@pytest.fixture()
def agents():
    from vision_inspection_system_amd import config as C
    from vision_inspection_system_amd.agents import VLMAuditorAgent, VLMInspectorAgent
    C.set_config(C.Config(vlm_inspector_provider="mock", vlm_auditor_provider="mock", max_image_dimension=2048))
    yield VLMInspectorAgent(), VLMAuditorAgent()
    C.set_config(None)


@pytest.mark.parametrize("case", VEC["encode"], ids=[c["name"] for c in VEC["encode"]])
def test_encode_image_optimized_is_byte_identical(case, agents, tmp_path):
    insp, aud = agents
    recipe = {r["name"]: r for r in ENCODE_RECIPES}.get(case["name"])
    if recipe is not None:
        path = tmp_path / (case["name"] + (".jpg" if recipe["format"] == "JPEG" else ".png"))
        make_recipe_image(recipe, path)
    else:       # a data file of the reference's own test suite (tests/conftest.py:19-59), carried as bytes
        path = tmp_path / "reference_1x1.jpg"
        path.write_bytes(base64.b64decode(case["file_base64"]))
    assert _digest(insp._encode_image_optimized, Path(path)) == case["inspector_default"]
    assert _digest(insp._encode_image_optimized, Path(path), 64) == case["inspector_max64"]
    assert _digest(aud._encode_image_optimized, Path(path)) == case["auditor_default"]
    assert _digest(aud._encode_image_optimized, Path(path), 300) == case["auditor_max300"]


@pytest.mark.parametrize("case", VEC["encode_large"], ids=[c["recipe"]["name"] for c in VEC["encode_large"]])
def test_quality_retry_and_size_refusal(case, agents, tmp_path):
    """> 5 MB at q85 -> re-encode at q60; the Inspector refuses > 10 MB (ValueError), the Auditor does not check."""
    insp, aud = agents
    assert case["recipe"] in LARGE_RECIPES
    path = tmp_path / "big.png"
    make_recipe_image(case["recipe"], path)
    assert _digest(insp._encode_image_optimized, Path(path), case["max_size"]) == case["inspector"]
    assert _digest(aud._encode_image_optimized, Path(path), case["max_size"]) == case["auditor"]


def test_auditor_keeps_the_reference_quirks(agents, tmp_path):
    """The Auditor copy has no 'LA' conversion: Pillow then refuses to write the JPEG (vlm_auditor.py:95-96)."""
    la = next(c for c in VEC["encode"] if c["name"] == "gray_alpha")
    assert la["auditor_default"] == {"error": "OSError"} and "sha256" in la["inspector_default"]


HOST = "/root/reference"


@pytest.mark.skipif(not os.path.exists(os.path.join(HOST, "utils", "prompts.py")),
                    reason="the host application (reference checkout) is not present on this machine")
def test_host_prompts_are_used_verbatim_and_format_identically(monkeypatch):
    """Drop-in deployment: with the host application's utils.prompts importable the product sends ITS prompt text
    (a4); the formatted prompts for three contexts hash to what the reference's own call produced."""
    import types
    pkg = types.ModuleType("utils")
    pkg.__path__ = [os.path.join(HOST, "utils")]          # namespace package: utils/__init__.py needs dotenv
    monkeypatch.setitem(sys.modules, "utils", pkg)
    monkeypatch.delitem(sys.modules, "utils.prompts", raising=False)
    import vision_inspection_system_amd.prompts as P
    P = importlib.reload(P)
    try:
        assert P.HOST_PROMPTS is True
        for name, text in (("inspector", P.INSPECTOR_PROMPT), ("auditor", P.AUDITOR_PROMPT)):
            b = text.encode("utf-8")
            assert {"sha256": hashlib.sha256(b).hexdigest(), "length": len(b)} == VEC["prompt_templates"][name]
        for rec in VEC["prompts"]:
            ctx = rec["context"]
            ins = P.INSPECTOR_PROMPT.format(criticality=ctx["criticality"], domain=ctx["domain"] or "general",
                                            user_notes=ctx["user_notes"] or "None provided")
            au = P.AUDITOR_PROMPT.format(criticality=ctx["criticality"], domain=ctx["domain"] or "general")
            for text, want in ((ins, rec["inspector"]), (au, rec["auditor"])):
                b = text.encode("utf-8")
                assert {"sha256": hashlib.sha256(b).hexdigest(), "length": len(b)} == want
    finally:
        monkeypatch.undo()
        sys.modules.pop("utils.prompts", None)
        importlib.reload(P)


def test_agent_messages_use_the_same_format_call(agents, tmp_path):
    """The agents' request builder formats the prompt with the reference's defaults ("general", "None provided") and
    sends [text, image_url] in that order (vlm_inspector.py:452-470, vlm_auditor.py:188-206)."""
    from vision_inspection_system_amd import prompts as P
    from vision_inspection_system_amd.schemas import InspectionContext
    insp, aud = agents
    path = tmp_path / "a.png"
    make_recipe_image(ENCODE_RECIPES[0], path)
    ctx = InspectionContext(image_id="i", criticality="high")
    m = insp._messages(Path(path), ctx)
    assert m[0]["content"][0]["text"] == P.INSPECTOR_PROMPT.format(criticality="high", domain="general",
                                                                   user_notes="None provided")
    m = aud._messages(Path(path), ctx)
    assert m[0]["content"][0]["text"] == P.AUDITOR_PROMPT.format(criticality="high", domain="general")
    assert [p["type"] for p in m[0]["content"]] == ["text", "image_url"]
