"""Row f1 (post-gather step): Inspector/Auditor agreement scoring, src/safety/consensus.py:18-151.

score = 0.4*[conditions agree] + 0.3*Jaccard(defect types) + 0.2*count agreement + 0.1*confidence agreement,
rounded to 4 decimals, >= 0.9999 -> 1.0, agree iff score >= 0.7.  Two clean reports only count as agreement
when BOTH are high-confidence (:60-70).  Parity vectors: tests/golden/reference_postprocess.json ("consensus").
"""
from __future__ import annotations

import logging

from .schemas import ConsensusResult, VLMAnalysisResult

logger = logging.getLogger("vision_inspection_system_amd.consensus")
_LEVEL = {"high": 3, "medium": 2, "low": 1}


def analyze_consensus(inspector_result: VLMAnalysisResult, auditor_result: VLMAnalysisResult) -> ConsensusResult:
    same_condition = inspector_result.overall_condition == auditor_result.overall_condition
    n_i, n_a = len(inspector_result.defects), len(auditor_result.defects)
    t_i, t_a = set(inspector_result.defect_types), set(auditor_result.defect_types)
    union = t_i | t_a
    type_score = len(t_i & t_a) / len(union) if union else 1.0

    if n_i == 0 and n_a == 0 and not (inspector_result.overall_confidence == "high"
                                      and auditor_result.overall_confidence == "high"):
        logger.warning("Both models report 'no defects' but confidence is not HIGH for both - treating as disagreement")
        type_score = 0.0
        same_condition = False

    diff = abs(n_i - n_a)
    count_score = 1.0 if diff <= 1 else max(0, 1 - (diff / max(n_i, n_a, 1)))
    conf_score = 1.0 - (abs(_LEVEL.get(inspector_result.overall_confidence, 2)
                            - _LEVEL.get(auditor_result.overall_confidence, 2)) / 2)
    score = 0.4 * (1.0 if same_condition else 0.0) + 0.3 * type_score + 0.2 * count_score + 0.1 * conf_score
    score = round(score, 4)
    if score >= 0.9999:
        score = 1.0
    agree = score >= 0.7

    details = None
    if not agree:
        parts = []
        if not same_condition:
            parts.append(f"Condition: Inspector says '{inspector_result.overall_condition}', "
                         f"Auditor says '{auditor_result.overall_condition}'")
        if n_i != n_a:
            parts.append(f"Count: Inspector found {n_i} defects, Auditor found {n_a}")
        only_i, only_a = t_i - t_a, t_a - t_i
        if only_i:
            parts.append(f"Inspector found: {', '.join(only_i)}")
        if only_a:
            parts.append(f"Auditor found: {', '.join(only_a)}")
        details = "; ".join(parts)
    return ConsensusResult(models_agree=agree, inspector_result=inspector_result, auditor_result=auditor_result,
                           agreement_score=score, disagreement_details=details)
