"""Boundary B3: LangGraph node functions of the hot path with the reference's signatures and state keys.

``run_inspector(state)`` / ``run_auditor(state)`` mirror src/orchestration/nodes.py:115-211 / :214-296:
build the ``InspectionContext`` from ``state["context"]``, obtain a fresh agent, at most one retry with
backoff (:40-47), store ``result.model_dump()`` under ``inspector_result`` / ``auditor_result``, record
failures in ``error`` / ``failure_history`` / ``has_critical_failure``, and upgrade the context criticality
when the Inspector infers a higher one (:188-206).  They are plain ``state -> state`` callables and can be
registered in a ``StateGraph(InspectionState)`` unchanged.
"""
from __future__ import annotations

import logging
import time
from pathlib import Path
from typing import Any, Dict, List, Union

from .agents import get_auditor, get_inspector
from .schemas import InspectionContext, VLMAnalysisResult

logger = logging.getLogger("vision_inspection_system_amd.workflow")

InspectionState = Dict[str, Any]  # the reference's TypedDict (src/orchestration/state.py:92-129) is a dict at run time
_sleep = time.sleep  # tests replace this


def _normalize_image_input(image_path: Union[str, List[str]]) -> List[str]:
    if isinstance(image_path, str):
        return [image_path]
    if isinstance(image_path, list):
        return image_path
    raise ValueError(f"Invalid image_path type: {type(image_path)}")


def _backoff_delay(retry_count: int) -> float:
    return min(2.0 ** retry_count, 10.0)


def _failed(reason_text: str, error_msg: str) -> VLMAnalysisResult:
    return VLMAnalysisResult(object_identified="unknown", overall_condition="uncertain", defects=[],
                             overall_confidence="low", analysis_reasoning=reason_text, analysis_failed=True,
                             failure_reason=error_msg)


def _run_with_retry(state: InspectionState, who: str, counter_key: str, call, max_retries: int = 1):
    retry = state.get(counter_key, 0)
    result = None
    while retry <= max_retries:
        try:
            if retry > 0:
                delay = _backoff_delay(retry - 1)
                logger.info(f"Retrying {who} (attempt {retry + 1}/{max_retries + 1}) after {delay:.1f}s delay...")
                _sleep(delay)
            result = call()
            if result.analysis_failed:
                raise Exception(result.failure_reason or (f"{who} analysis failed" if who == "Inspector" else f"{who} verification failed"))
            break
        except Exception as e:
            logger.warning(f"{who} attempt {retry + 1} failed: {e}")
            if retry < max_retries:
                retry += 1
                state[counter_key] = retry
                continue
            error_msg = f"{who} failed after {retry + 1} attempt(s): {str(e)}"
            state["error"] = error_msg
            state["failure_history"] = state.get("failure_history", []) + [error_msg]
            state["has_critical_failure"] = True
            verb = "Analysis" if who == "Inspector" else "Verification"
            result = _failed(f"{verb} failed after retries: {str(e)}", error_msg)
            break
    return result


def run_inspector(state: InspectionState) -> InspectionState:
    """Inspector node (src/orchestration/nodes.py:115-211)."""
    state["current_step"] = "inspector_analysis"
    context = InspectionContext(**state["context"])
    inspector = get_inspector()
    image_path = Path(_normalize_image_input(state["image_path"])[0])
    result = _run_with_retry(state, "Inspector", "inspector_retry_count",
                             lambda: inspector.analyze(image_path, context))
    if result:
        state["inspector_result"] = result.model_dump()
        if result.inferred_criticality:
            user, inferred = context.criticality, result.inferred_criticality
            order = {"low": 0, "medium": 1, "high": 2}
            if user != inferred and order.get(inferred, 1) > order.get(user, 1):
                logger.warning(f"Upgrading criticality from '{user}' to '{inferred}' based on agent analysis")
                state["context"]["criticality"] = inferred
                state["context"]["criticality_upgraded"] = True
                state["context"]["original_criticality"] = user
                state["context"]["upgrade_reason"] = result.inferred_criticality_reasoning
        if not result.analysis_failed:
            logger.info(f"Inspector found {len(result.defects)} defects")
    return state


def run_auditor(state: InspectionState) -> InspectionState:
    """Auditor node (src/orchestration/nodes.py:214-296)."""
    state["current_step"] = "auditor_verification"
    context = InspectionContext(**state["context"])
    inspector_result = VLMAnalysisResult(**state["inspector_result"])
    auditor = get_auditor()
    image_path = Path(_normalize_image_input(state["image_path"])[0])
    result = _run_with_retry(state, "Auditor", "auditor_retry_count",
                             lambda: auditor.verify(image_path, context, inspector_result))
    if result:
        state["auditor_result"] = result.model_dump()
        if not result.analysis_failed:
            logger.info(f"Auditor found {len(result.defects)} defects")
    return state
