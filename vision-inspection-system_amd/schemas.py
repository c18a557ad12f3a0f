"""Output contract of the hot path (row a8): pydantic v2 models with the reference's field names,
literals, defaults and validators (src/schemas/models.py:11-131,:134-270).

If the host application's own ``src.schemas.models`` is importable (drop-in deployment inside the
reference tree) those classes are re-exported instead, so objects crossing the boundary are the
application's own types.
"""
from __future__ import annotations

import time
from datetime import datetime
from typing import Any, Dict, List, Literal, Optional

from pydantic import BaseModel, Field, field_validator, model_validator

try:  # pragma: no cover - only inside the reference application
    from src.schemas.models import (BoundingBox, ConsensusResult, DefectInfo, InspectionContext,  # type: ignore
                                    SafetyVerdict, VLMAnalysisResult)
    HOST_SCHEMAS = True
except Exception:
    HOST_SCHEMAS = False

if not HOST_SCHEMAS:

    class BoundingBox(BaseModel):
        """Box in PERCENT of the image (0-100), origin top-left (src/schemas/models.py:11-54)."""
        x: float = Field(...)
        y: float = Field(...)
        width: float = Field(...)
        height: float = Field(...)

        @field_validator("x", "y", "width", "height")
        @classmethod
        def _non_negative(cls, v):
            if v < 0:
                raise ValueError("Coordinates must be non-negative")
            return v

        @model_validator(mode="after")
        def _percent_range(self):
            if not (0 <= self.x <= 100):
                raise ValueError(f"X coordinate must be between 0 and 100, got {self.x}")
            if not (0 <= self.y <= 100):
                raise ValueError(f"Y coordinate must be between 0 and 100, got {self.y}")
            if self.width <= 0 or self.width > 100:
                raise ValueError(f"Width must be between 0 and 100, got {self.width}")
            if self.height <= 0 or self.height > 100:
                raise ValueError(f"Height must be between 0 and 100, got {self.height}")
            if self.x + self.width > 100:
                raise ValueError(f"Bounding box exceeds image width: x={self.x}, width={self.width}")
            if self.y + self.height > 100:
                raise ValueError(f"Bounding box exceeds image height: y={self.y}, height={self.height}")
            return self

        def is_reasonable(self, min_area_percent: float = 0.1, max_area_percent: float = 50.0) -> bool:
            area = (self.width * self.height) / 100.0
            return min_area_percent <= area <= max_area_percent

    class DefectInfo(BaseModel):
        """One detected defect (src/schemas/models.py:57-82)."""
        defect_id: str = Field(default_factory=lambda: f"defect_{int(time.time() * 1000)}")
        type: str = Field(...)
        location: str = Field(...)
        bbox: Optional[BoundingBox] = None
        safety_impact: Literal["CRITICAL", "MODERATE", "COSMETIC"] = Field(...)
        reasoning: str = Field(...)
        confidence: Literal["high", "medium", "low"] = Field(...)
        recommended_action: str = Field(...)

        @field_validator("type")
        @classmethod
        def _normalise_type(cls, v: str) -> str:
            return v.lower().strip()

        def is_critical(self) -> bool:
            return self.safety_impact == "CRITICAL"

    class VLMAnalysisResult(BaseModel):
        """What one agent returns for one image (src/schemas/models.py:85-131)."""
        object_identified: str = Field(...)
        overall_condition: Literal["damaged", "good", "uncertain"] = Field(...)
        defects: List[DefectInfo] = Field(default_factory=list)
        overall_confidence: Literal["high", "medium", "low"] = Field(...)
        analysis_reasoning: Optional[str] = None
        inferred_criticality: Optional[Literal["low", "medium", "high"]] = None
        inferred_criticality_reasoning: Optional[str] = None
        analysis_failed: bool = False
        failure_reason: Optional[str] = None
        timestamp: datetime = Field(default_factory=datetime.utcnow)

        @property
        def has_defects(self) -> bool:
            return len(self.defects) > 0

        @property
        def critical_defect_count(self) -> int:
            return sum(1 for d in self.defects if d.is_critical())

        @property
        def defect_types(self) -> List[str]:
            return list(set(d.type for d in self.defects))

    _SEMANTIC_GROUPS = (
        {"crack", "hairline_crack", "fracture", "fissure"},
        {"rust", "corrosion", "oxidation"},
        {"scratch", "scrape", "abrasion"},
        {"dent", "deformation"},
        {"discoloration", "stain"},
    )

    def _same_kind(a: DefectInfo, b: DefectInfo) -> bool:
        ta, tb = a.type.lower().strip(), b.type.lower().strip()
        return ta == tb or any(ta in g and tb in g for g in _SEMANTIC_GROUPS)

    def _iou_at_least(b1: Optional[BoundingBox], b2: Optional[BoundingBox], thr: float = 0.5) -> bool:
        if b1 is None or b2 is None:
            return False
        ix0, iy0 = max(b1.x, b2.x), max(b1.y, b2.y)
        ix1, iy1 = min(b1.x + b1.width, b2.x + b2.width), min(b1.y + b1.height, b2.y + b2.height)
        if ix1 <= ix0 or iy1 <= iy0:
            return False
        inter = (ix1 - ix0) * (iy1 - iy0)
        union = b1.width * b1.height + b2.width * b2.height - inter
        return union != 0 and inter / union >= thr

    class ConsensusResult(BaseModel):
        """Inspector/Auditor agreement record; ``combined_defects`` is recomputed on every construction
        (src/schemas/models.py:134-241): an auditor defect of the same semantic kind whose box overlaps
        an inspector defect with IoU >= 0.5 is merged into it, everything else is kept from both."""
        models_agree: bool = Field(...)
        inspector_result: VLMAnalysisResult
        auditor_result: VLMAnalysisResult
        agreement_score: float = Field(..., ge=0, le=1)
        disagreement_details: Optional[str] = None
        combined_defects: List[DefectInfo] = Field(default_factory=list)

        @model_validator(mode="after")
        def compute_combined_defects(self):
            aud = list(self.auditor_result.defects)
            used = [False] * len(aud)
            merged: List[DefectInfo] = []
            for d in self.inspector_result.defects:
                for i, a in enumerate(aud):
                    if not used[i] and _same_kind(d, a) and _iou_at_least(d.bbox, a.bbox):
                        used[i] = True
                        break
                merged.append(d)
            merged.extend(a for i, a in enumerate(aud) if not used[i])
            self.combined_defects = merged
            return self

    class SafetyVerdict(BaseModel):
        """Final verdict record (src/schemas/models.py:244-261)."""
        verdict: Literal["SAFE", "UNSAFE", "REQUIRES_HUMAN_REVIEW"] = Field(...)
        reason: str = Field(...)
        requires_human: bool = Field(...)
        confidence_level: Literal["high", "medium", "low"] = Field(...)
        triggered_gates: List[str] = Field(default_factory=list)
        defect_summary: Dict[str, Any] = Field(default_factory=dict)
        errors: List[str] = Field(default_factory=list)
        timestamp: datetime = Field(default_factory=datetime.utcnow)

    class InspectionContext(BaseModel):
        """Per-image request context (src/schemas/models.py:264-270); unknown keys are ignored."""
        image_id: str
        criticality: Literal["low", "medium", "high"] = "medium"
        domain: Optional[str] = None
        reference_standards: Optional[List[str]] = None
        user_notes: Optional[str] = None


__all__ = ["BoundingBox", "DefectInfo", "VLMAnalysisResult", "ConsensusResult", "SafetyVerdict",
           "InspectionContext", "HOST_SCHEMAS"]
