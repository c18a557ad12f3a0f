"""Row f1 (post-gather step): deterministic safety gates over a ConsensusResult.

Behavioural restatement of ``SafetyGateEngine.evaluate`` (src/safety/gates.py:133-775): a defect
pre-filter (:186-287), nine gates that are ALL evaluated and recorded (:165-645) where the first failing
gate in order fixes the verdict, and a verdict resolution for the "nothing blocked" cases (:647-775).
Parity is pinned on the structured outcome - verdict, requires_human, confidence_level, triggered_gates,
per-gate (id, passed), defect counts - by tests/golden/reference_postprocess.json ("gates", captured from
the reference's code, not from its stale test expectations).  Explanatory strings
are this module's own wording.

Domain rules are configuration data: the host application's ``config/safety_rules.yaml`` when present
(``VIS_SAFETY_RULES`` or ./config/safety_rules.yaml), else the table below (same values as the
reference ships, config/safety_rules.yaml:25-62).
"""
from __future__ import annotations

import logging
import os
from typing import Any, Dict, List, Optional, Tuple

from .config import get_config
from .schemas import ConsensusResult, DefectInfo, InspectionContext, SafetyVerdict

logger = logging.getLogger("vision_inspection_system_amd.safety")

GATE_ERROR_STATE = "GATE_0_ERROR_STATE"
GATE_CRITICAL_DEFECT = "GATE_1_CRITICAL_DEFECT"
GATE_DOMAIN_ZERO_TOLERANCE = "GATE_2_DOMAIN_ZERO_TOLERANCE"
GATE_MODEL_DISAGREEMENT = "GATE_3_MODEL_DISAGREEMENT"
GATE_LOW_CONFIDENCE = "GATE_4_LOW_CONFIDENCE"
GATE_DEFECT_COUNT = "GATE_5_DEFECT_COUNT"
GATE_HIGH_CRITICALITY = "GATE_6_HIGH_CRITICALITY"
GATE_NO_DEFECTS = "GATE_7_NO_DEFECTS"
GATE_AUDITOR_UNCERTAIN = "GATE_8_AUDITOR_UNCERTAIN"
GATE_DEFAULT_CONSERVATIVE = "GATE_DEFAULT_CONSERVATIVE"

GATE_DISPLAY_NAMES = {
    GATE_ERROR_STATE: "Error State Check", GATE_CRITICAL_DEFECT: "Critical Defect Check",
    GATE_DOMAIN_ZERO_TOLERANCE: "Domain Zero Tolerance", GATE_MODEL_DISAGREEMENT: "Model Agreement Check",
    GATE_LOW_CONFIDENCE: "Confidence Threshold", GATE_DEFECT_COUNT: "Defect Count Limit",
    GATE_HIGH_CRITICALITY: "High Criticality Check", GATE_NO_DEFECTS: "No Defects Verification",
    GATE_AUDITOR_UNCERTAIN: "Auditor Certainty Check", GATE_DEFAULT_CONSERVATIVE: "Conservative Fallback",
}

DEFAULT_DOMAINS: Dict[str, Dict[str, Any]] = {
    "aerospace": {"zero_tolerance_types": ["crack", "fatigue", "delamination"], "require_human_review_always": True},
    "medical": {"zero_tolerance_types": ["contamination", "crack", "corrosion"], "require_human_review_always": True},
    "automotive": {"zero_tolerance_types": ["crack", "fracture", "structural_damage"],
                   "require_human_review_always": False},
    "food_processing": {"zero_tolerance_types": ["contamination", "foreign_object", "mold", "discoloration"],
                        "require_human_review_always": False},
    "general": {"zero_tolerance_types": [], "require_human_review_always": False},
}


def load_domain_rules() -> Dict[str, Dict[str, Any]]:
    for path in (os.environ.get("VIS_SAFETY_RULES"), os.path.join("config", "safety_rules.yaml")):
        if path and os.path.exists(path):
            try:
                import yaml
                with open(path) as f:
                    doc = yaml.safe_load(f) or {}
                if isinstance(doc.get("domains"), dict):
                    return doc["domains"]
            except Exception as e:  # pragma: no cover
                logger.warning(f"Could not load {path}: {e}")
    return DEFAULT_DOMAINS


def _bbox_in_bounds(d: DefectInfo) -> bool:
    b = d.bbox
    return not (b.x < 0 or b.x > 100 or b.y < 0 or b.y > 100 or b.width <= 0 or b.width > 100
                or b.height <= 0 or b.height > 100 or b.x + b.width > 100 or b.y + b.height > 100)


def _screen_defects(consensus: ConsensusResult, context: InspectionContext) -> List[DefectInfo]:
    """Pre-filter of gates.py:186-287: geometry sanity, low-confidence drop (unless high criticality) and the
    four false-positive heuristics for non-critical findings."""
    ins, aud = consensus.inspector_result, consensus.auditor_result
    n_i, n_a = len(ins.defects), len(aud.defects)
    one_sided = (n_i == 0) != (n_a == 0)
    very_low_agreement = consensus.agreement_score < 0.4
    ins_clean = n_i == 0 and ins.overall_confidence == "high" and ins.overall_condition == "good"
    aud_clean = n_a == 0 and aud.overall_confidence == "high" and aud.overall_condition == "good"
    both_good = ins.overall_condition == "good" and aud.overall_condition == "good"
    both_good_confident = (both_good and ins.overall_confidence in ("high", "medium")
                           and aud.overall_confidence in ("high", "medium") and (n_i > 0 or n_a > 0))
    kept: List[DefectInfo] = []
    for d in consensus.combined_defects:
        if d.bbox:
            if not _bbox_in_bounds(d):
                continue
            area = (d.bbox.width * d.bbox.height) / 100.0
            if area < 0.05 or area > 50.0:
                continue
        if d.confidence == "low" and context.criticality != "high":
            continue
        non_critical = d.safety_impact in ("MODERATE", "COSMETIC", "MINOR")
        if non_critical:
            if ins_clean or aud_clean:
                continue
            if both_good_confident and d.confidence != "high":
                continue
        if very_low_agreement and one_sided and non_critical:
            if ins_clean or aud_clean or (both_good and d.confidence in ("low", "medium")):
                continue
        kept.append(d)
    return kept


def _flagged_for_domain(defect_type: str, rules: Dict[str, Any]) -> bool:
    t = defect_type.lower()
    return any(z.lower() in t or t in z.lower() for z in rules.get("zero_tolerance_types", []))


class SafetyGateEngine:
    def __init__(self, domains: Optional[Dict[str, Dict[str, Any]]] = None, settings=None):
        self.domains = domains if domains is not None else load_domain_rules()
        self.settings = settings or get_config()

    def _domain_rules(self, domain: Optional[str]) -> Dict[str, Any]:
        if domain and domain.lower() in self.domains:
            return self.domains[domain.lower()]
        return self.domains.get("general", {})

    def evaluate(self, consensus: ConsensusResult, context: InspectionContext) -> SafetyVerdict:
        ins, aud = consensus.inspector_result, consensus.auditor_result
        max_auto = getattr(self.settings, "max_defects_auto", 2)
        high_needs_review = getattr(self.settings, "high_criticality_requires_review", True)
        gates: List[Dict[str, Any]] = []
        triggered: List[str] = []
        decision: Optional[Tuple[str, str, str, bool]] = None

        def record(gate_id: str, passed: bool, message: str, details: Optional[dict] = None) -> None:
            gates.append({"gate_id": gate_id, "display_name": GATE_DISPLAY_NAMES.get(gate_id, gate_id),
                          "passed": passed, "message": message, "details": details or {}})

        def block(gate_id: str, verdict: str, reason: str, confidence: str, human: bool = False) -> None:
            nonlocal decision
            if decision is None:  # only the first failing gate decides; later ones are still recorded
                triggered.append(gate_id)
                decision = (verdict, reason, confidence, human)

        errors = []
        if ins.analysis_failed:
            errors.append(f"Inspector: {ins.failure_reason or 'Analysis failed'}")
        if aud.analysis_failed:
            errors.append(f"Auditor: {aud.failure_reason or 'Analysis failed'}")

        # gate 0: a failed analysis can never yield a pass
        record(GATE_ERROR_STATE, not errors, "No analysis errors" if not errors else f"{len(errors)} analysis error(s)",
               {"errors": errors} if errors else {})
        if errors:
            block(GATE_ERROR_STATE, "UNSAFE", "Analysis failed: " + "; ".join(errors), "low", True)

        defects = _screen_defects(consensus, context)
        n = len(defects)
        critical = [d for d in defects if d.safety_impact == "CRITICAL"]
        moderate = [d for d in defects if d.safety_impact == "MODERATE"]
        cosmetic = [d for d in defects if d.safety_impact == "COSMETIC"]
        i_conf, a_conf, a_cond = ins.overall_confidence, aud.overall_confidence, aud.overall_condition
        rules = self._domain_rules(context.domain)
        agree_conf = "high" if consensus.models_agree else "medium"

        # gate 1: critical defects
        shaky = bool(critical) and consensus.agreement_score < 0.5 and not consensus.models_agree
        record(GATE_CRITICAL_DEFECT, not critical, f"{len(critical) or 'No'} critical defects",
               {"critical_count": len(critical), "types": [d.type for d in critical], "low_agreement_warning": shaky})
        if critical:
            kinds = ", ".join(d.type for d in critical)
            if shaky:
                block(GATE_CRITICAL_DEFECT, "UNSAFE",
                      f"Critical defect(s) reported ({kinds}) while the models strongly disagree "
                      f"(agreement {consensus.agreement_score:.0%}); conservative automatic UNSAFE.", "medium")
            else:
                block(GATE_CRITICAL_DEFECT, "UNSAFE", f"{len(critical)} critical safety defect(s) detected: {kinds}",
                      agree_conf)

        # gate 2: domain zero-tolerance types
        flagged = [d for d in defects if _flagged_for_domain(d.type, rules)]
        g2_ok = not (flagged and rules.get("require_human_review_always", False))
        record(GATE_DOMAIN_ZERO_TOLERANCE, g2_ok, "Passed" if g2_ok else f"{len(flagged)} domain violations",
               {"domain": context.domain, "flagged": [d.type for d in flagged]})
        if not g2_ok:
            block(GATE_DOMAIN_ZERO_TOLERANCE, "UNSAFE",
                  f"Zero-tolerance defect type(s) for domain '{context.domain}': "
                  f"{', '.join(d.type for d in flagged)}", "high")

        # gate 3: the two models must agree
        record(GATE_MODEL_DISAGREEMENT, consensus.models_agree, f"Agreement: {consensus.agreement_score:.0%}",
               {"agreement_score": consensus.agreement_score, "models_agree": consensus.models_agree})
        if not consensus.models_agree:
            if n > 0:
                block(GATE_MODEL_DISAGREEMENT, "UNSAFE",
                      f"Models disagree and defects remain after screening. {consensus.disagreement_details}.", "medium")
            else:
                block(GATE_MODEL_DISAGREEMENT, "SAFE",
                      f"Models disagree but no defect survived screening. {consensus.disagreement_details}.", "medium")

        # gate 4: neither model may be low-confidence
        low_conf = i_conf == "low" or a_conf == "low"
        record(GATE_LOW_CONFIDENCE, not low_conf, f"Inspector: {i_conf}, Auditor: {a_conf}",
               {"inspector_confidence": i_conf, "auditor_confidence": a_conf})
        if low_conf:
            block(GATE_LOW_CONFIDENCE, "UNSAFE" if n > 0 else "SAFE",
                  f"Low model confidence (Inspector: {i_conf}, Auditor: {a_conf}) with {n} defect(s).", "low")

        # gate 5: too many defects for an automatic pass
        record(GATE_DEFECT_COUNT, n <= max_auto, f"{n} defects (limit: {max_auto})",
               {"defect_count": n, "limit": max_auto})
        if n > max_auto:
            block(GATE_DEFECT_COUNT, "UNSAFE", f"{n} defects exceed the automatic limit of {max_auto}.", "medium")

        # gate 6: high-criticality parts
        high = context.criticality == "high"
        if high and n == 0:
            g6_ok = i_conf == "high" and a_conf == "high"
            g6_msg = ("High criticality, no defects, both models HIGH confidence" if g6_ok else
                      f"High criticality, no defects, insufficient confidence (Inspector: {i_conf}, Auditor: {a_conf})")
        else:
            g6_ok = not (high and n > 0 and high_needs_review)
            g6_msg = f"Criticality: {context.criticality}, Defects: {n}"
        record(GATE_HIGH_CRITICALITY, g6_ok, g6_msg,
               {"criticality": context.criticality, "defect_count": n, "inspector_confidence": i_conf,
                "auditor_confidence": a_conf})
        if not g6_ok:
            if high and n == 0:
                block(GATE_HIGH_CRITICALITY, "SAFE",
                      f"High-criticality part without defects but confidence is not HIGH on both models "
                      f"(Inspector: {i_conf}, Auditor: {a_conf}).", "medium")
            else:
                block(GATE_HIGH_CRITICALITY, "UNSAFE", f"High-criticality part with {n} defect(s).", "high")

        # gate 7: verified-clean check (recorded only; it decides when nothing blocked)
        bad_boxes = [d.type for d in consensus.combined_defects if d.bbox and not _bbox_in_bounds(d)]
        both_high = i_conf == "high" and a_conf == "high"
        high_agreement = consensus.agreement_score > 0.8
        g7_ok = n == 0 and not bad_boxes and both_high and high_agreement and not errors
        if g7_ok:
            g7_msg = "No defects, both models HIGH confidence, high agreement, no errors - verified clean"
        elif n == 0:
            missing = []
            if bad_boxes:
                missing.append(f"invalid bbox coordinates: {', '.join(bad_boxes)}")
            if not both_high:
                missing.append(f"both models HIGH confidence (Inspector: {i_conf}, Auditor: {a_conf})")
            if not high_agreement:
                missing.append(f"agreement > 0.8 (score {consensus.agreement_score:.2f})")
            if errors:
                missing.append("no analysis errors")
            g7_msg = "No defects but missing: " + ", ".join(missing)
        else:
            g7_msg = f"{n} valid defects found"
        record(GATE_NO_DEFECTS, g7_ok, g7_msg,
               {"defect_count": n, "has_invalid_bboxes": bool(bad_boxes), "invalid_bbox_defects": bad_boxes,
                "inspector_confidence": i_conf, "auditor_confidence": a_conf, "both_high_confidence": both_high,
                "agreement_score": consensus.agreement_score, "high_agreement": high_agreement,
                "no_errors": not errors})

        # gate 8: the auditor itself must be certain
        unsure = a_cond == "uncertain" or a_conf == "low"
        record(GATE_AUDITOR_UNCERTAIN, not unsure, f"Auditor condition: {a_cond}, confidence: {a_conf}",
               {"auditor_condition": a_cond, "auditor_confidence": a_conf})
        if unsure:
            block(GATE_AUDITOR_UNCERTAIN, "UNSAFE" if n > 0 else "SAFE",
                  f"Auditor uncertain (condition: {a_cond}, confidence: {a_conf}) with {n} defect(s).", "low")

        def verdict(v, reason, conf, human, summary):
            summary["all_gate_results"] = gates
            return SafetyVerdict(verdict=v, reason=reason, requires_human=human, confidence_level=conf,
                                 triggered_gates=triggered, errors=errors, defect_summary=summary)

        if decision is None and g7_ok:
            triggered.append(GATE_NO_DEFECTS)
            return verdict("SAFE", "No defects reported by either model; every gate passed with HIGH confidence.",
                           "high", False, {"total_defects": 0, "verification_passed": True})
        if decision is not None:
            v, reason, conf, human = decision
            return verdict(v, reason, conf, human, {"total_defects": n, "critical": len(critical),
                                                    "moderate": len(moderate), "cosmetic": len(cosmetic)})
        # nothing blocked, not verified clean: only non-critical findings are left
        if not critical and not moderate and cosmetic:
            if high:
                triggered.append(GATE_DEFAULT_CONSERVATIVE)
                record(GATE_DEFAULT_CONSERVATIVE, False,
                       f"High criticality with {len(cosmetic)} cosmetic defects - cosmetic only, SAFE",
                       {"criticality": context.criticality, "cosmetic_count": len(cosmetic)})
                reason = f"High-criticality part with {len(cosmetic)} cosmetic defect(s) only - no safety impact."
            else:
                triggered.append(GATE_NO_DEFECTS)
                reason = f"Only cosmetic defects detected ({len(cosmetic)}). No safety impact."
            return verdict("SAFE", reason, agree_conf, False, {"total_defects": n, "cosmetic": len(cosmetic)})
        triggered.append(GATE_DEFAULT_CONSERVATIVE)
        record(GATE_DEFAULT_CONSERVATIVE, False,
               f"Conservative: {len(moderate)} moderate, {len(cosmetic)} cosmetic defects",
               {"moderate": len(moderate), "cosmetic": len(cosmetic)})
        what = f"{len(moderate)} MODERATE" if moderate else f"{n} unclassified"
        kinds = ", ".join(d.type for d in defects[:3]) + ("..." if len(defects) > 3 else "")
        return verdict("UNSAFE", f"Defects detected: {what} defect(s). Types: {kinds}", agree_conf, False,
                       {"total_defects": n, "moderate": len(moderate), "cosmetic": len(cosmetic),
                        "defect_types": [d.type for d in defects]})


def evaluate_safety(consensus: ConsensusResult, context: InspectionContext) -> SafetyVerdict:
    return SafetyGateEngine().evaluate(consensus, context)
