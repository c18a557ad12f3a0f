"""Row a10 + section 8(e): the per-image workflow and the batch entry points, sharded data-parallel.

* ``run_inspection`` - one image through the hot-path nodes (Inspector -> Auditor -> consensus -> safety
  gates when available), starting from the reference's initial-state literal (src/orchestration/graph.py:162-189).
  UI / explanation / database / PDF nodes of the reference graph are out of scope.
* ``run_multi_image_inspection`` - the reference's batch loop (graph.py:269-387), same arguments and the same
  return shape ``{session_id, image_results, session_results, processing_time}``.  The reference walks the
  images sequentially; here, when ``torch.distributed`` is initialised with W > 1 ranks (one rank per GPU,
  backend "nccl" = RCCL over xGMI, or "gloo" on CPU for tests), rank r inspects ``image_paths[r::W]`` - whole
  images, full model replica per GPU, no tensor exchanged - and the per-image result records are exchanged
  with ONE collective: all_gather of a length header + all_gather of padded UTF-8 JSON bytes.  Every rank
  returns the full, input-ordered result, aggregated once with ``aggregate_session_results``.
* ``run_batch_inspection`` - the README name (README.md:154-160) for the same thing.
A failing image (or a rank whose engine raised) yields ``completed=False`` records; the batch never crashes
(graph.py:349-357).
"""
from __future__ import annotations

import json
import logging
import os
import time
import uuid
from datetime import datetime
from typing import Any, Dict, List, Optional

from .aggregation import aggregate_session_results
from .consensus import analyze_consensus
from .nodes import run_auditor, run_inspector
from .schemas import InspectionContext, VLMAnalysisResult

logger = logging.getLogger("vision_inspection_system_amd.batch")


def _local_batching() -> bool:
    """True when both agents are served by the local engine (so requests can share one decode loop)."""
    from .config import LOCAL_PROVIDER, get_config
    cfg = get_config()
    return getattr(cfg, "vlm_inspector_provider", "") == LOCAL_PROVIDER and \
        getattr(cfg, "vlm_auditor_provider", "") == LOCAL_PROVIDER and os.environ.get("VIS_BATCH_DECODE", "1") != "0"


# ----------------------------------------------------------------------------- collective
def _dist():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist
    except Exception:
        pass
    return None


def gather_records(records: List[dict], world: Optional[int] = None) -> List[dict]:
    """All ranks contribute a list of JSON-serialisable records; every rank receives the concatenation in
    rank order.  One length all_gather + one padded-bytes all_gather (RCCL when the group is nccl)."""
    dist = _dist()
    if dist is None or dist.get_world_size() == 1:
        return list(records)
    import torch
    backend = dist.get_backend()
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    payload = json.dumps(records, default=str).encode("utf-8")
    W = dist.get_world_size()
    n = torch.tensor([len(payload)], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(W)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    cap = max(max(sizes), 1)
    buf = torch.zeros(cap, dtype=torch.uint8, device=dev)
    if payload:
        buf[:len(payload)] = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(dev)
    outs = [torch.zeros(cap, dtype=torch.uint8, device=dev) for _ in range(W)]
    dist.all_gather(outs, buf)
    merged: List[dict] = []
    for o, sz in zip(outs, sizes):
        if sz:
            merged.extend(json.loads(bytes(o[:sz].cpu().tolist()).decode("utf-8")))
    return merged


# ----------------------------------------------------------------------------- single image
def _jsonable(x):
    return json.loads(json.dumps(x, default=str)) if x is not None else None


def run_inspection(image_path: str, criticality: str = "medium", domain: Optional[str] = None,
                   user_notes: Optional[str] = None) -> Dict[str, Any]:
    """One image through Inspector -> Auditor -> consensus (-> safety gates)."""
    thread_id = str(uuid.uuid4())[:8]
    state: Dict[str, Any] = {
        "image_path": image_path,
        "context": {"image_id": str(uuid.uuid4())[:8], "criticality": criticality, "domain": domain,
                    "user_notes": user_notes},
        "request_id": thread_id, "start_time": time.time(),
        "inspector_result": None, "auditor_result": None, "consensus": None, "safety_verdict": None,
        "clean_verification": None, "requires_human_review": False, "human_decision": None, "human_notes": None,
        "explanation": None, "report_path": None, "processing_time": None, "error": None,
        "failure_history": [], "has_critical_failure": False, "inspector_retry_count": 0,
        "auditor_retry_count": 0, "current_step": "pending",
    }
    state = run_inspector(state)
    state = run_auditor(state)
    return _finish_state(state)


def _finish_state(state: Dict[str, Any]) -> Dict[str, Any]:
    """consensus -> safety gates -> bookkeeping, shared by the single-image and the batched flow."""
    state["current_step"] = "consensus_analysis"
    inspector = VLMAnalysisResult(**state["inspector_result"])
    auditor = VLMAnalysisResult(**state["auditor_result"])
    consensus = analyze_consensus(inspector, auditor)
    state["consensus"] = consensus.model_dump()
    try:
        from .gates import evaluate_safety
        verdict = evaluate_safety(consensus, InspectionContext(**state["context"]))
        state["safety_verdict"] = verdict.model_dump()
        state["requires_human_review"] = verdict.requires_human
    except ImportError:
        state["safety_verdict"] = None
    state["current_step"] = "completed"
    state["processing_time"] = time.time() - state["start_time"]
    return state


def run_inspections_batched(image_paths: List[str], criticality: str = "medium", domain: Optional[str] = None,
                            user_notes: Optional[str] = None) -> List[Dict[str, Any]]:
    """Several images through Inspector -> Auditor with ONE shared decode loop per agent (local provider):
    the weights are streamed once per generated token for all images in flight.  Returns one state dict per
    image, same keys as ``run_inspection``.  An image whose batched analysis failed is retried once through the
    single-image node (the reference's node-level retry, nodes.py:138-181)."""
    from .agents import get_auditor, get_inspector
    states = []
    for p in image_paths:
        states.append({"image_path": p,
                       "context": {"image_id": str(uuid.uuid4())[:8], "criticality": criticality, "domain": domain,
                                   "user_notes": user_notes},
                       "request_id": str(uuid.uuid4())[:8], "start_time": time.time(), "inspector_result": None,
                       "auditor_result": None, "consensus": None, "safety_verdict": None, "clean_verification": None,
                       "requires_human_review": False, "human_decision": None, "human_notes": None,
                       "explanation": None, "report_path": None, "processing_time": None, "error": None,
                       "failure_history": [], "has_critical_failure": False, "inspector_retry_count": 0,
                       "auditor_retry_count": 0, "current_step": "pending"})
    contexts = [InspectionContext(**s["context"]) for s in states]
    for res, st in zip(get_inspector().analyze_many(image_paths, contexts), states):
        if res.analysis_failed:
            st["inspector_retry_count"] = 1      # the batched attempt was attempt 1; the node makes the final one
            run_inspector(st)
        else:
            st["current_step"] = "inspector_analysis"
            st["inspector_result"] = res.model_dump()
    for res, st in zip(get_auditor().verify_many(image_paths, contexts), states):
        if res.analysis_failed:
            st["auditor_retry_count"] = 1
            run_auditor(st)
        else:
            st["current_step"] = "auditor_verification"
            st["auditor_result"] = res.model_dump()
    return [_finish_state(st) for st in states]


def _image_record(image_id: str, image_path: str, result: Dict[str, Any]) -> Dict[str, Any]:
    return {
        "image_id": image_id, "image_path": image_path,
        "inspector_result": _jsonable(result.get("inspector_result")),
        "auditor_result": _jsonable(result.get("auditor_result")),
        "consensus": _jsonable(result.get("consensus")),
        "safety_verdict": _jsonable(result.get("safety_verdict")),
        "clean_verification": result.get("clean_verification"),
        "explanation": result.get("explanation"),
        "decision_support": result.get("decision_support", {}),
        "report_path": result.get("report_path"),
        "processing_time": result.get("processing_time", 0),
        "error": result.get("error"),
        "failure_history": result.get("failure_history", []),
        "completed": True,
    }


# ----------------------------------------------------------------------------- batch
def run_multi_image_inspection(image_paths: List[str], criticality: str = "medium", domain: Optional[str] = None,
                               user_notes: Optional[str] = None, session_id: Optional[str] = None,
                               image_id_map: Optional[Dict[str, str]] = None,
                               _inspect=run_inspection) -> Dict[str, Any]:
    start = datetime.now()
    dist = _dist()
    rank = dist.get_rank() if dist else 0
    world = dist.get_world_size() if dist else 1
    if not session_id:
        session_id = str(uuid.uuid4())[:8]
    if world > 1:  # ids must agree across ranks: rank 0's choice wins
        box = [session_id]
        dist.broadcast_object_list(box, src=0)
        session_id = box[0]

    mine: List[dict] = []
    my_idx = list(range(rank, len(image_paths), world))
    pre: Dict[int, Dict[str, Any]] = {}
    if _inspect is run_inspection and len(my_idx) > 1 and _local_batching():
        try:  # shared-decode fast path; any problem falls back to the per-image loop below
            group = int(os.environ.get("VIS_MAX_BATCH", "64"))
            for g0 in range(0, len(my_idx), group):
                chunk = my_idx[g0:g0 + group]
                outs = run_inspections_batched([image_paths[i] for i in chunk], criticality, domain, user_notes)
                pre.update(dict(zip(chunk, outs)))
        except Exception as e:
            logger.error(f"batched inspection failed ({e}); falling back to per-image processing", exc_info=True)
            pre = {}
    for idx in my_idx:
        image_path = image_paths[idx]
        image_id = image_id_map[image_path] if image_id_map and image_path in image_id_map \
            else f"{session_id}-{idx:04d}"
        try:
            result = pre[idx] if idx in pre else _inspect(image_path=image_path, criticality=criticality,
                                                              domain=domain, user_notes=user_notes)
            rec = _image_record(image_id, image_path, result)
        except Exception as e:
            logger.error(f"Failed to process image {image_path}: {e}", exc_info=True)
            rec = {"image_id": image_id, "image_path": image_path, "error": str(e), "failure_history": [str(e)],
                   "completed": False}
        rec["_index"] = idx
        rec["_rank"] = rank
        mine.append(rec)

    records = gather_records(mine, world)
    records.sort(key=lambda r: r["_index"])
    image_results: Dict[str, Dict[str, Any]] = {}
    verdicts: List[str] = []
    for rec in records:
        rec.pop("_index", None)
        rec.pop("_rank", None)
        image_id = rec.pop("image_id")
        image_results[image_id] = rec
        if rec.get("completed"):
            verdicts.append((rec.get("safety_verdict") or {}).get("verdict", "UNKNOWN"))
    raw = aggregate_session_results({k: {**v, "safety_verdict": v.get("safety_verdict") or {},
                                         "consensus": v.get("consensus") or {}} for k, v in image_results.items()})
    end = datetime.now()
    duration = (end - start).total_seconds()
    session_results = {**raw, "session_id": session_id, "session_duration": duration,
                       "session_start_time": start.isoformat(), "session_end_time": end.isoformat(),
                       "per_image_verdicts": verdicts}
    return {"session_id": session_id, "image_results": image_results, "session_results": session_results,
            "processing_time": duration}


def run_batch_inspection(image_paths: List[str], criticality: str = "medium", domain: Optional[str] = None,
                         **kwargs) -> Dict[str, Any]:
    """README.md:154-160 signature; returns exactly what ``run_multi_image_inspection`` returns."""
    return run_multi_image_inspection(image_paths, criticality=criticality, domain=domain, **kwargs)
