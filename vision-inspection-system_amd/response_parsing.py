"""Rows a6 / a7 of the hot path: model text -> dict -> validated dict.

Behavioural restatement of
  * ``_parse_json_robust``       src/agents/vlm_inspector.py:142-267 (Auditor copy: vlm_auditor.py:236-326,
                                 which has NO partial-result rescue branch), and
  * ``_validate_and_fix_result`` src/agents/vlm_inspector.py:269-431 (Auditor copy: vlm_auditor.py:328-472).
Parity is pinned by tests/golden/reference_postprocess.json (vectors captured from the reference's own
code).  Quirks are reproduced on purpose - e.g. a defect whose ``reasoning`` is ``null`` raises here
exactly as it does in the reference, which turns the whole analysis into ``analysis_failed=True``
(SURVEY.md appendix A).
"""
from __future__ import annotations

import json
import logging
import re
from typing import Any, Dict, Optional

_FENCE = re.compile(r"```(?:json)?\s*([\s\S]*?)```")
_REASONING_ESCAPED = re.compile(r'"analysis_reasoning"\s*:\s*"([^"]*(?:\\.[^"]*)*)"', re.DOTALL)
_REASONING_PLAIN = re.compile(r'"analysis_reasoning"\s*:\s*"([^"]*)"')
_OBJECT = re.compile(r'"object_identified"\s*:\s*"([^"]*)"')

_VAGUE_REASONING = ("possible", "might be", "appears to be", "could be", "uncertain", "unclear")
_VAGUE_LOCATION = ("somewhere", "various", "multiple", "general", "areas")
_IMPACTS = ("CRITICAL", "MODERATE", "COSMETIC")
_CONFIDENCES = ("high", "medium", "low")

_null_logger = logging.getLogger("vision_inspection_system_amd.parse")


def _longest_balanced_object(s: str) -> Optional[str]:
    """Longest substring that starts at some '{', ends at its matching '}' and parses as JSON.

    Every '{' is tried as a start (O(n^2) like the reference); quote / backslash state is tracked from
    the start position, a backslash skips the next character whether or not we are inside a string.
    """
    best, best_len = None, 0
    n = len(s)
    for start in range(n):
        if s[start] != "{":
            continue
        depth, in_string, skip = 0, False, False
        for j in range(start, n):
            ch = s[j]
            if skip:
                skip = False
                continue
            if ch == "\\":
                skip = True
                continue
            if ch == '"':
                in_string = not in_string
                continue
            if in_string:
                continue
            if ch == "{":
                depth += 1
            elif ch == "}":
                depth -= 1
                if depth == 0:
                    cand = s[start:j + 1]
                    if len(cand) > best_len:
                        try:
                            json.loads(cand)
                            best, best_len = cand, len(cand)
                        except json.JSONDecodeError:
                            pass
                    break
    return best


def parse_json_robust(text: str, rescue_partial: bool = True, logger: Optional[logging.Logger] = None) -> Dict[str, Any]:
    """Extract the JSON object of a model reply.

    Order: first parsable fenced block -> longest balanced object -> first '{' .. last '}' ->
    (Inspector only, ``rescue_partial``) a partial dict rebuilt from the ``analysis_reasoning`` string ->
    ``ValueError``.
    """
    log = logger or _null_logger
    text = text.strip()
    for block in _FENCE.findall(text):
        try:
            return json.loads(block.strip())
        except json.JSONDecodeError:
            continue
    cand = _longest_balanced_object(text)
    if cand:
        try:
            return json.loads(cand)
        except json.JSONDecodeError:
            pass
    lo, hi = text.find("{"), text.rfind("}") + 1
    if lo != -1 and hi > lo:
        try:
            return json.loads(text[lo:hi])
        except json.JSONDecodeError:
            pass
    if rescue_partial and "analysis_reasoning" in text:
        m = _REASONING_ESCAPED.search(text) or _REASONING_PLAIN.search(text)
        reasoning = m.group(1).replace('\\"', '"').replace("\\n", "\n") if m else None
        if reasoning:
            log.warning("JSON parsing failed but extracted analysis_reasoning - returning partial result")
            partial: Dict[str, Any] = {"analysis_reasoning": reasoning}
            om = _OBJECT.search(text)
            partial["object_identified"] = om.group(1) if om else "unknown"
            partial["defects"] = []
            partial["overall_condition"] = "uncertain"
            partial["overall_confidence"] = "low"
            return partial
    log.error(f"JSON parsing failed. Raw text (first 500 chars): {text[:500]}")
    raise ValueError("Failed to parse JSON from model response")


def validate_and_fix_result(result: Dict[str, Any], logger: Optional[logging.Logger] = None,
                            who: str = "") -> Dict[str, Any]:
    """Fill defaults, boost clean-image confidence, sanitise defects and boxes (mutates and returns ``result``)."""
    log = logger or _null_logger
    for key, default in (("object_identified", "unknown"), ("overall_condition", "uncertain"),
                         ("overall_confidence", "low")):
        if key not in result:
            result[key] = default
    if "defects" not in result:
        result["defects"] = []

    # a clean verdict ("good", no defects) is nudged one confidence level up
    if len(result.get("defects", [])) == 0 and result.get("overall_condition", "uncertain") == "good":
        conf = result.get("overall_confidence", "low")
        if conf == "low":
            result["overall_confidence"] = "medium"
            log.info(f"Boosted {who}confidence from 'low' to 'medium' for clean image")
        elif conf == "medium":
            result["overall_confidence"] = "high"
            log.info(f"Boosted {who}confidence from 'medium' to 'high' for clean image")

    kept = []
    for defect in result.get("defects", []):
        if not isinstance(defect, dict):
            continue
        defect.setdefault("type", "unspecified")
        defect.setdefault("location", "unspecified")
        defect.setdefault("safety_impact", "MODERATE")
        defect.setdefault("reasoning", "No reasoning provided")
        defect.setdefault("confidence", "low")
        defect.setdefault("recommended_action", "Further inspection recommended")
        if defect["safety_impact"] not in _IMPACTS:
            defect["safety_impact"] = "MODERATE"
        if defect["confidence"] not in _CONFIDENCES:
            defect["confidence"] = "low"

        defect.get("type", "").lower()  # the reference lower-cases the type here (raises on null, unused otherwise)
        confidence = defect.get("confidence", "low")
        reasoning = defect.get("reasoning", "").lower()
        if confidence == "low" and any(v in reasoning for v in _VAGUE_REASONING):
            log.warning(f"Filtering out {who}low-confidence defect with vague reasoning: {defect.get('type')} - "
                        f"'{defect.get('reasoning', '')[:50]}'")
            continue

        if "bbox" in defect and defect["bbox"]:
            bbox = defect["bbox"]
            if isinstance(bbox, dict) and all(k in bbox for k in ("x", "y", "width", "height")):
                x, y = bbox.get("x", 0), bbox.get("y", 0)
                w, h = bbox.get("width", 0), bbox.get("height", 0)
                if any(v > 100 for v in (x, y, w, h) if v > 0):
                    # looks like pixels; without the model's input size it cannot be converted
                    log.warning(f"Bbox values > 100 detected, assuming pixel format: {bbox}")
                    defect["bbox"] = None
                    defect["bbox_approximate"] = True
                elif x < 0 or x > 100 or y < 0 or y > 100 or w <= 0 or w > 100 or h <= 0 or h > 100:
                    log.warning(f"Bbox values out of valid percentage range (0-100): {bbox}")
                    defect["bbox"] = None
                    defect["bbox_approximate"] = True
                elif x + w > 100 or y + h > 100:
                    log.warning(f"Bbox exceeds image bounds: x+width={x + w}, y+height={y + h}")
                    defect["bbox"] = None
                    defect["bbox_approximate"] = True
                else:
                    area = (w * h) / 100.0  # percent of the image
                    if area < 0.05:
                        log.warning(f"Bbox very small (area={area:.2f}% < 0.05%) - may be noise: {bbox}")
                        if confidence == "low" and area < 0.02:
                            log.warning(f"Filtering out {who}very low-confidence defect with extremely tiny bbox: "
                                        f"{defect.get('type')}")
                            continue
                        defect["bbox_approximate"] = True
                    elif area > 50.0:
                        log.warning(f"Bbox too large (area={area:.2f}% > 50%) - likely error: {bbox}")
                        defect["bbox"] = None
                        defect["bbox_approximate"] = True
                    else:
                        defect["bbox"] = {"x": max(0, min(100, x)), "y": max(0, min(100, y)),
                                          "width": max(0.1, min(100, w)), "height": max(0.1, min(100, h))}
            else:
                defect["bbox"] = None

        if not defect.get("bbox") and confidence == "low":
            location = defect.get("location", "").lower()
            if any(v in location for v in _VAGUE_LOCATION):
                log.warning(f"Filtering out {who}low-confidence defect with no bbox and vague location: "
                            f"{defect.get('type')}")
                continue
        kept.append(defect)

    result["defects"] = kept
    return result
