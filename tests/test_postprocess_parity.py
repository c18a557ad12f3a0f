"""Rows a6/a7/a8 + f1: product post-processing vs vectors captured from the reference's own code
(tests/golden/gen_reference_postprocess.py -> reference_postprocess.json).  Exact equality."""
import copy
import json
import os

import pytest

from helpers import GOLDEN

VEC = json.load(open(os.path.join(GOLDEN, "reference_postprocess.json")))


def _run(parse, validate, text):
    from vision_inspection_system_amd.schemas import VLMAnalysisResult
    out = {}
    try:
        parsed = parse(text)
        out["parsed"] = copy.deepcopy(parsed)
    except Exception as e:
        out["parse_error"] = type(e).__name__
        return out
    try:
        fixed = validate(parsed)
        out["validated"] = copy.deepcopy(fixed)
    except Exception as e:
        out["validate_error"] = type(e).__name__
        return out
    try:
        dumped = json.loads(VLMAnalysisResult(**fixed).model_dump_json())
        dumped.pop("timestamp", None)
        for d in dumped.get("defects", []):
            d.pop("defect_id", None)
        out["model"] = dumped
    except Exception as e:
        out["model_error"] = type(e).__name__
    return out


@pytest.mark.parametrize("case", VEC["postprocess"], ids=[c["name"] for c in VEC["postprocess"]])
def test_parse_validate_matches_reference(case):
    from vision_inspection_system_amd.response_parsing import parse_json_robust, validate_and_fix_result
    got_i = _run(lambda t: parse_json_robust(t, rescue_partial=True), validate_and_fix_result, case["text"])
    got_a = _run(lambda t: parse_json_robust(t, rescue_partial=False),
                 lambda d: validate_and_fix_result(d, who="auditor "), case["text"])
    assert json.loads(json.dumps(got_i)) == case["inspector"]
    assert json.loads(json.dumps(got_a)) == case["auditor"]


def test_vector_coverage():
    names = {c["name"] for c in VEC["postprocess"]}
    assert len(names) >= 25
    insp = {c["name"]: c["inspector"] for c in VEC["postprocess"]}
    aud = {c["name"]: c["auditor"] for c in VEC["postprocess"]}
    # the rescue branch exists only in the Inspector
    assert "parsed" in insp["truncated_with_reasoning"] and aud["truncated_with_reasoning"].get("parse_error") == "ValueError"
    assert insp["garbage"].get("parse_error") == "ValueError"
    assert "validate_error" in insp["reasoning_null_crash"]


@pytest.mark.parametrize("case", VEC["consensus"], ids=[c["name"] for c in VEC["consensus"]])
def test_consensus_matches_reference(case):
    from vision_inspection_system_amd.consensus import analyze_consensus
    from vision_inspection_system_amd.schemas import VLMAnalysisResult
    cons = analyze_consensus(VLMAnalysisResult(**copy.deepcopy(case["inspector"])),
                             VLMAnalysisResult(**copy.deepcopy(case["auditor"])))
    assert cons.agreement_score == case["agreement_score"]
    assert cons.models_agree == case["models_agree"]
    assert [d.type for d in cons.combined_defects] == case["combined_types"]
    # set iteration order is hash dependent: compare the detail clauses as sets of words
    got, exp = cons.disagreement_details, case["disagreement_details"]
    assert (got is None) == (exp is None)
    if exp:
        norm = lambda s: sorted(sorted(part.replace(",", " ").split()) for part in s.split("; "))
        assert norm(got) == norm(exp)


@pytest.mark.parametrize("case", VEC["aggregate"], ids=[c["name"] for c in VEC["aggregate"]])
def test_aggregate_matches_reference(case):
    from vision_inspection_system_amd.aggregation import aggregate_session_results
    assert aggregate_session_results(copy.deepcopy(case["image_results"])) == case["expected"]


def test_reference_test_suite_cases():
    """The live cases of the reference's tests/test_safety_gates.py:26-112,:295-341 (consensus + schemas)."""
    from vision_inspection_system_amd.consensus import analyze_consensus
    from vision_inspection_system_amd.schemas import DefectInfo, VLMAnalysisResult
    clean = dict(object_identified="bolt", overall_condition="good", defects=[], overall_confidence="high")
    c = analyze_consensus(VLMAnalysisResult(**clean), VLMAnalysisResult(**clean))
    assert c.models_agree and c.agreement_score >= 0.9
    d1 = dict(type="crack", location="a", safety_impact="CRITICAL", reasoning="r", confidence="high", recommended_action="x")
    d2 = dict(type="rust", location="b", safety_impact="MODERATE", reasoning="r", confidence="medium", recommended_action="y")
    dam = dict(object_identified="bolt", overall_condition="damaged", defects=[d1], overall_confidence="high")
    c2 = analyze_consensus(VLMAnalysisResult(**dam), VLMAnalysisResult(**clean))
    assert not c2.models_agree and "Condition" in c2.disagreement_details
    both = analyze_consensus(VLMAnalysisResult(**dam),
                             VLMAnalysisResult(**{**dam, "defects": [d2]}))
    assert len(both.combined_defects) == 2
    assert DefectInfo(**{**d1, "type": "  CRACK "}).type == "crack"
    r = VLMAnalysisResult(**{**dam, "defects": [d1, d2, {**d1, "type": "Crack"}]})
    assert r.critical_defect_count == 2 and sorted(r.defect_types) == ["crack", "rust"]


@pytest.mark.parametrize("case", VEC["gates"], ids=[f"{c['name']}-{c['context']['criticality']}-{c['context'].get('domain','')}" for c in VEC["gates"]])
def test_gates_match_reference(case):
    """Structured outcome of the safety gates vs the reference's code (not its stale tests)."""
    from vision_inspection_system_amd import config as C
    from vision_inspection_system_amd.consensus import analyze_consensus
    from vision_inspection_system_amd.gates import DEFAULT_DOMAINS, SafetyGateEngine
    from vision_inspection_system_amd.schemas import InspectionContext, VLMAnalysisResult
    by_name = {c["name"]: c for c in VEC["consensus"]}[case["name"]]
    cons = analyze_consensus(VLMAnalysisResult(**copy.deepcopy(by_name["inspector"])),
                             VLMAnalysisResult(**copy.deepcopy(by_name["auditor"])))
    got = SafetyGateEngine(domains=DEFAULT_DOMAINS, settings=C.Config()).evaluate(cons, InspectionContext(**case["context"]))
    exp = case["verdict"]
    assert got.verdict == exp["verdict"]
    assert got.requires_human == exp["requires_human"]
    assert got.confidence_level == exp["confidence_level"]
    assert got.triggered_gates == exp["triggered_gates"]
    assert got.errors == exp["errors"]
    g_got = [(g["gate_id"], g["passed"], g["details"]) for g in got.defect_summary["all_gate_results"]]
    g_exp = [(g["gate_id"], g["passed"], g["details"]) for g in exp["defect_summary"]["all_gate_results"]]
    assert g_got == g_exp
    for k, v in exp["defect_summary"].items():
        if k != "all_gate_results":
            assert got.defect_summary[k] == v, k


def test_reference_gate_suite_cases():
    """The live cases of the reference's tests/test_safety_gates.py:115-292 with the CODE's answers
    (its test_gate_3_model_disagreement expectation is stale: the code returns SAFE/UNSAFE, never review)."""
    from vision_inspection_system_amd import config as C
    from vision_inspection_system_amd.consensus import analyze_consensus
    from vision_inspection_system_amd.gates import DEFAULT_DOMAINS, SafetyGateEngine
    from vision_inspection_system_amd.schemas import InspectionContext, VLMAnalysisResult
    eng = SafetyGateEngine(domains=DEFAULT_DOMAINS, settings=C.Config())
    d = dict(type="crack", location="a", safety_impact="CRITICAL", reasoning="r", confidence="high", recommended_action="x")
    clean = dict(object_identified="bolt", overall_condition="good", defects=[], overall_confidence="high")
    dam = dict(object_identified="bolt", overall_condition="damaged", defects=[d], overall_confidence="high")
    ctx = InspectionContext(image_id="t", criticality="medium")
    v = eng.evaluate(analyze_consensus(VLMAnalysisResult(**dam), VLMAnalysisResult(**dam)), ctx)
    assert v.verdict == "UNSAFE" and "GATE_1_CRITICAL_DEFECT" in v.triggered_gates
    v = eng.evaluate(analyze_consensus(VLMAnalysisResult(**clean), VLMAnalysisResult(**clean)), ctx)
    assert v.verdict == "SAFE" and "GATE_7_NO_DEFECTS" in v.triggered_gates
    assert len(v.defect_summary["all_gate_results"]) >= 7
    cos = {**d, "safety_impact": "COSMETIC", "type": "scratch"}
    c2 = dict(object_identified="bolt", overall_condition="damaged", defects=[cos], overall_confidence="high")
    v = eng.evaluate(analyze_consensus(VLMAnalysisResult(**c2), VLMAnalysisResult(**c2)), ctx)
    assert v.verdict == "SAFE"
    v = eng.evaluate(analyze_consensus(VLMAnalysisResult(**clean), VLMAnalysisResult(**{**clean, "overall_condition": "damaged"})), ctx)
    assert v.verdict == "SAFE" and v.requires_human is False and v.triggered_gates == ["GATE_3_MODEL_DISAGREEMENT"]
