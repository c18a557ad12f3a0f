"""DEV-ONLY generator of the post-processing golden vectors (G1-G4 of SURVEY.md section 8c).

Imports the reference's OWN deterministic code from /root/reference (read-only) and records its
outputs; only the resulting JSON (inputs + expected outputs) is committed.  Third-party packages the
reference needs but this image lacks (dotenv, pydantic_settings, colorlog, langchain_core, langgraph, cv2 ...)
are never reached: the package ``__init__`` files that pull them in are bypassed with namespace
packages, ``utils.config`` is replaced by a namespace carrying the handful of fields the hot path reads
and ``utils.logger`` by a stdlib logger factory - the reference's own files
(src/agents/vlm_inspector.py, vlm_auditor.py, src/safety/consensus.py, gates.py,
src/orchestration/session_aggregation.py, src/schemas/models.py) run unmodified.

Usage:  python tests/golden/gen_reference_postprocess.py     (writes reference_postprocess.json)
"""
import copy
import json
import logging
import os
import sys
import types

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _ns_package(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m
    return m


def load_reference():
    sys.path.insert(0, REF)
    _ns_package("utils", os.path.join(REF, "utils"))
    _ns_package("src", os.path.join(REF, "src"))
    _ns_package("src.agents", os.path.join(REF, "src", "agents"))
    _ns_package("src.safety", os.path.join(REF, "src", "safety"))
    _ns_package("src.orchestration", os.path.join(REF, "src", "orchestration"))
    cfgmod = types.ModuleType("utils.config")
    cfgmod.config = types.SimpleNamespace(
        huggingface_api_key="hf_dummy", groq_api_key=None, log_level="ERROR",
        vlm_inspector_model="Qwen/Qwen2.5-VL-7B-Instruct", vlm_inspector_temperature=0.1,
        vlm_inspector_max_tokens=2048, vlm_inspector_provider="huggingface",
        vlm_auditor_model="Qwen/Qwen2.5-VL-7B-Instruct", vlm_auditor_temperature=0.2,
        vlm_auditor_max_tokens=2048, vlm_auditor_provider="huggingface", max_image_dimension=2048,
        max_defects_auto=2, high_criticality_requires_review=True)
    sys.modules["utils.config"] = cfgmod
    logmod = types.ModuleType("utils.logger")
    logging.basicConfig(level=logging.CRITICAL)
    logmod.setup_logger = lambda name, **kw: logging.getLogger("ref." + str(name))
    logmod.set_request_id = lambda *_a, **_k: None
    sys.modules["utils.logger"] = logmod
    for name in ("langchain_core", "langchain_core.messages", "langchain_core.language_models"):
        m = types.ModuleType(name)
        m.HumanMessage = m.SystemMessage = m.BaseChatModel = object
        sys.modules[name] = m
    from src.agents.vlm_inspector import VLMInspectorAgent
    from src.agents.vlm_auditor import VLMAuditorAgent
    from src.safety.consensus import analyze_consensus
    from src.safety.gates import evaluate_safety
    from src.orchestration.session_aggregation import aggregate_session_results
    from src.schemas.models import VLMAnalysisResult, InspectionContext
    return (VLMInspectorAgent(), VLMAuditorAgent(), analyze_consensus, evaluate_safety, aggregate_session_results,
            VLMAnalysisResult, InspectionContext)


def defect(**kw):
    d = {"type": "crack", "location": "upper left edge", "bbox": {"x": 10, "y": 20, "width": 15, "height": 10},
         "safety_impact": "CRITICAL", "reasoning": "Visible fracture line across the weld", "confidence": "high",
         "recommended_action": "Replace"}
    d.update(kw)
    return d


def reply(defects=None, **kw):
    r = {"object_identified": "steel bracket", "overall_condition": "damaged", "defects": defects or [],
         "overall_confidence": "high", "analysis_reasoning": "Bracket shows damage."}
    r.update(kw)
    return r


def raw_text_cases():
    J = json.dumps
    base = reply([defect()])
    c = []
    c.append(("plain_json", J(base)))
    c.append(("fenced_json", "Here is the result:\n```json\n" + J(base) + "\n```\nLet me know."))
    c.append(("fenced_no_lang", "```\n" + J(reply([], overall_condition="good")) + "\n```"))
    c.append(("two_fences_first_bad", "```json\n{not json}\n```\nthen\n```json\n" + J(base) + "\n```"))
    c.append(("prose_then_json", "Sure! The analysis follows. " + J(base) + " Hope this helps {really}."))
    c.append(("nested_braces_in_string", J(reply([defect(reasoning="brace } inside { text")]))))
    c.append(("two_objects_longest_wins", J({"a": 1}) + " and " + J(base)))
    c.append(("truncated_with_reasoning", '{"object_identified": "gear", "analysis_reasoning": "Teeth look \\"worn\\" overall\\nsecond line", "defects": [{"type": "wear", '))
    c.append(("truncated_no_reasoning", '{"object_identified": "gear", "defects": [{"type": "wear", '))
    c.append(("garbage", "I cannot analyse this image."))
    c.append(("empty", ""))
    c.append(("list_reply", "```json\n[1, 2, 3]\n```"))
    c.append(("missing_fields", J({"defects": [{"type": "Rust "}]})))
    c.append(("clean_low_conf_boost", J(reply([], overall_condition="good", overall_confidence="low"))))
    c.append(("clean_medium_conf_boost", J(reply([], overall_condition="good", overall_confidence="medium"))))
    c.append(("clean_high_conf_kept", J(reply([], overall_condition="good", overall_confidence="high"))))
    c.append(("uncertain_no_boost", J(reply([], overall_condition="uncertain", overall_confidence="low"))))
    c.append(("invalid_enums", J(reply([defect(safety_impact="SEVERE", confidence="certain")]))))
    c.append(("vague_low_conf_dropped", J(reply([defect(confidence="low", reasoning="This might be a crack")]))))
    c.append(("vague_high_conf_kept", J(reply([defect(confidence="high", reasoning="This might be a crack")]))))
    c.append(("pixel_bbox", J(reply([defect(bbox={"x": 320, "y": 200, "width": 50, "height": 40})]))))
    c.append(("bbox_overflow", J(reply([defect(bbox={"x": 90, "y": 20, "width": 15, "height": 10})]))))
    c.append(("bbox_negative", J(reply([defect(bbox={"x": -5, "y": 20, "width": 15, "height": 10})]))))
    c.append(("bbox_zero_width", J(reply([defect(bbox={"x": 5, "y": 20, "width": 0, "height": 10})]))))
    c.append(("bbox_tiny_flagged", J(reply([defect(bbox={"x": 5, "y": 5, "width": 0.2, "height": 0.2})]))))
    c.append(("bbox_tiny_low_conf_dropped", J(reply([defect(confidence="low", bbox={"x": 5, "y": 5, "width": 0.1, "height": 0.1})]))))
    c.append(("bbox_tiny_low_conf_kept", J(reply([defect(confidence="low", bbox={"x": 5, "y": 5, "width": 0.2, "height": 0.2})]))))
    c.append(("bbox_huge", J(reply([defect(bbox={"x": 0, "y": 0, "width": 90, "height": 80})]))))
    c.append(("bbox_missing_key", J(reply([defect(bbox={"x": 1, "y": 2, "width": 3})]))))
    c.append(("bbox_not_dict", J(reply([defect(bbox=[1, 2, 3, 4])]))))
    c.append(("bbox_null", J(reply([defect(bbox=None)]))))
    c.append(("bbox_float_clamp", J(reply([defect(bbox={"x": 0.0, "y": 99.5, "width": 100, "height": 0.5})]))))
    c.append(("vague_location_dropped", J(reply([defect(confidence="low", bbox=None, location="Various areas of the part")]))))
    c.append(("vague_location_medium_kept", J(reply([defect(confidence="medium", bbox=None, location="Various areas")]))))
    c.append(("non_dict_defect_skipped", J(reply([defect(), "oops", 7]))))
    c.append(("reasoning_null_crash", J(reply([defect(reasoning=None)]))))
    c.append(("inferred_criticality", J(reply([defect()], inferred_criticality="high", inferred_criticality_reasoning="load bearing"))))
    c.append(("bad_condition_literal", J(reply([], overall_condition="broken"))))
    c.append(("type_uppercase", J(reply([defect(type="  Hairline_Crack ")]))))
    c.append(("multiple_mixed", J(reply([defect(), defect(type="rust", safety_impact="MODERATE", confidence="medium", bbox={"x": 50, "y": 50, "width": 20, "height": 20}), defect(type="scratch", safety_impact="COSMETIC", confidence="low", reasoning="could be a scratch", bbox=None)]))))
    c.append(("bbox_string_values", J(reply([defect(bbox={"x": "10", "y": 20, "width": 15, "height": 10})]))))
    return c


def run_agent_case(agent, VLMAnalysisResult, text):
    out = {}
    try:
        parsed = agent._parse_json_robust(text)
        out["parsed"] = copy.deepcopy(parsed)
    except Exception as e:
        out["parse_error"] = type(e).__name__
        return out
    try:
        fixed = agent._validate_and_fix_result(parsed)
        out["validated"] = copy.deepcopy(fixed)
    except Exception as e:
        out["validate_error"] = type(e).__name__
        return out
    try:
        model = VLMAnalysisResult(**fixed)
        dumped = json.loads(model.model_dump_json())
        dumped.pop("timestamp", None)
        for d in dumped.get("defects", []):
            d.pop("defect_id", None)
        out["model"] = dumped
    except Exception as e:
        out["model_error"] = type(e).__name__
    return out


def result_dict(condition="damaged", confidence="high", defects=(), failed=False):
    d = {"object_identified": "bracket", "overall_condition": condition, "defects": [dict(x) for x in defects],
         "overall_confidence": confidence, "analysis_reasoning": "r"}
    if failed:
        d["analysis_failed"] = True
        d["failure_reason"] = "boom"
    return d


def consensus_cases():
    crack = defect()
    crack_moved = defect(bbox={"x": 60, "y": 60, "width": 15, "height": 10})
    crack_overlap = defect(type="fracture", bbox={"x": 11, "y": 21, "width": 15, "height": 10})
    rust = defect(type="rust", safety_impact="MODERATE", confidence="medium", bbox={"x": 40, "y": 40, "width": 10, "height": 10})
    scratch = defect(type="scratch", safety_impact="COSMETIC", confidence="high", bbox={"x": 70, "y": 10, "width": 5, "height": 5})
    dent_low = defect(type="dent", safety_impact="MODERATE", confidence="low", bbox=None)
    many = [defect(type=f"pit_{i}", safety_impact="MODERATE", bbox={"x": 5 * i, "y": 5, "width": 4, "height": 4}) for i in range(5)]
    C = []
    C.append(("agree_clean_high", result_dict("good", "high"), result_dict("good", "high")))
    C.append(("clean_but_medium", result_dict("good", "medium"), result_dict("good", "high")))
    C.append(("clean_both_low", result_dict("good", "low"), result_dict("good", "low")))
    C.append(("condition_disagree", result_dict("damaged", "high", [crack]), result_dict("good", "high")))
    C.append(("same_defect_overlap", result_dict("damaged", "high", [crack]), result_dict("damaged", "high", [crack_overlap])))
    C.append(("same_type_moved", result_dict("damaged", "high", [crack]), result_dict("damaged", "medium", [crack_moved])))
    C.append(("different_types", result_dict("damaged", "high", [crack]), result_dict("damaged", "high", [rust])))
    C.append(("count_gap", result_dict("damaged", "high", many), result_dict("damaged", "low", [crack])))
    C.append(("two_vs_two", result_dict("damaged", "high", [crack, rust]), result_dict("damaged", "high", [rust, scratch])))
    C.append(("cosmetic_only", result_dict("damaged", "high", [scratch]), result_dict("damaged", "high", [scratch])))
    C.append(("uncertain_auditor", result_dict("damaged", "high", [rust]), result_dict("uncertain", "low", [rust])))
    C.append(("inspector_failed", result_dict("uncertain", "low", [], failed=True), result_dict("good", "high")))
    C.append(("low_conf_defect", result_dict("damaged", "medium", [dent_low]), result_dict("damaged", "medium", [dent_low])))
    C.append(("moderate_pair", result_dict("damaged", "high", [rust]), result_dict("damaged", "high", [rust])))
    C.append(("many_agree", result_dict("damaged", "high", many), result_dict("damaged", "high", many)))
    return C


def main():
    (insp, aud, analyze_consensus, evaluate_safety, aggregate, VLMAnalysisResult, InspectionContext) = load_reference()
    out = {"postprocess": [], "consensus": [], "gates": [], "aggregate": []}
    for name, text in raw_text_cases():
        out["postprocess"].append({"name": name, "text": text,
                                   "inspector": run_agent_case(insp, VLMAnalysisResult, text),
                                   "auditor": run_agent_case(aud, VLMAnalysisResult, text)})
    contexts = [{"image_id": "img1", "criticality": "low"}, {"image_id": "img1", "criticality": "medium"},
                {"image_id": "img1", "criticality": "high", "domain": "aerospace"},
                {"image_id": "img1", "criticality": "medium", "domain": "mechanical_fasteners"}]
    per_image = {}
    for name, a, b in consensus_cases():
        ra, rb = VLMAnalysisResult(**copy.deepcopy(a)), VLMAnalysisResult(**copy.deepcopy(b))
        cons = analyze_consensus(ra, rb)
        out["consensus"].append({
            "name": name, "inspector": a, "auditor": b, "agreement_score": cons.agreement_score,
            "models_agree": cons.models_agree, "disagreement_details": cons.disagreement_details,
            "combined_types": [d.type for d in cons.combined_defects]})
        for ctx in contexts:
            v = evaluate_safety(cons, InspectionContext(**ctx))
            vd = json.loads(v.model_dump_json())
            vd.pop("timestamp", None)
            out["gates"].append({"name": name, "context": ctx, "verdict": vd})
            if ctx["criticality"] == "medium" and "domain" not in ctx:
                cd = json.loads(cons.model_dump_json())
                per_image[name] = {"completed": True, "safety_verdict": vd,
                                   "consensus": {"combined_defects": cd["combined_defects"]}}
    sessions = {
        "empty": {},
        "all": per_image,
        "safe_only": {k: v for k, v in per_image.items() if v["safety_verdict"]["verdict"] == "SAFE"},
        "with_failure": {**{k: per_image[k] for k in list(per_image)[:3]}, "broken": {"completed": False, "error": "x"}},
        "only_failed": {"broken": {"completed": False}},
    }
    for name, sess in sessions.items():
        out["aggregate"].append({"name": name, "image_results": sess, "expected": aggregate(copy.deepcopy(sess))})
    with open(os.path.join(HERE, "reference_postprocess.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print({k: len(v) for k, v in out.items()})
    from collections import Counter
    print(Counter(g["verdict"]["verdict"] for g in out["gates"]))


if __name__ == "__main__":
    main()
