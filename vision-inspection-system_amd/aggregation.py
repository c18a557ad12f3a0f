"""Row f1: session-level roll-up of per-image results, run once after the gather
(src/orchestration/session_aggregation.py:13-118).  Conservative verdict: any UNSAFE -> UNSAFE, else any
REQUIRES_HUMAN_REVIEW -> that, else all SAFE -> SAFE, else review."""
from __future__ import annotations

from typing import Any, Dict, List


def determine_aggregate_verdict(verdicts: List[str], total_defects: int) -> str:
    if not verdicts:
        return "UNKNOWN"
    if any(v == "UNSAFE" for v in verdicts):
        return "UNSAFE"
    if any(v == "REQUIRES_HUMAN_REVIEW" for v in verdicts):
        return "REQUIRES_HUMAN_REVIEW"
    if all(v == "SAFE" for v in verdicts):
        return "SAFE"
    return "REQUIRES_HUMAN_REVIEW"


def aggregate_session_results(image_results: Dict[str, Dict[str, Any]]) -> Dict[str, Any]:
    if not image_results:
        return {"total_images": 0, "completed_images": 0, "failed_images": 0, "aggregate_verdict": "UNKNOWN",
                "total_defects": 0, "critical_defects": 0, "moderate_defects": 0, "cosmetic_defects": 0}
    done = failed = total = 0
    by_impact = {"CRITICAL": 0, "MODERATE": 0, "COSMETIC": 0}
    verdicts: List[str] = []
    for result in image_results.values():
        if not result.get("completed", False):
            failed += 1
            continue
        done += 1
        verdicts.append(result.get("safety_verdict", {}).get("verdict", "UNKNOWN"))
        defects = result.get("consensus", {}).get("combined_defects", [])
        total += len(defects)
        for d in defects:
            impact = d.get("safety_impact", "COSMETIC")
            if impact in by_impact:
                by_impact[impact] += 1
    return {
        "total_images": len(image_results), "completed_images": done, "failed_images": failed,
        "aggregate_verdict": determine_aggregate_verdict(verdicts, total), "total_defects": total,
        "critical_defects": by_impact["CRITICAL"], "moderate_defects": by_impact["MODERATE"],
        "cosmetic_defects": by_impact["COSMETIC"],
        "verdict_distribution": {k: sum(1 for v in verdicts if v == k)
                                 for k in ("SAFE", "UNSAFE", "REQUIRES_HUMAN_REVIEW")},
    }
