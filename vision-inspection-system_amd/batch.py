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
    """True when at least one agent is served by the local engine: its requests then share one decode loop
    (``complete_many``); an agent on another provider is simply called request by request inside ``_many``."""
    from .config import LOCAL_PROVIDER, get_config
    cfg = get_config()
    local = LOCAL_PROVIDER in (getattr(cfg, "vlm_inspector_provider", ""), getattr(cfg, "vlm_auditor_provider", ""))
    return local and os.environ.get("VIS_BATCH_DECODE", "1") != "0"


# ----------------------------------------------------------------------------- collective
def _dist():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist
    except Exception:
        pass
    return None


class RankFailure(RuntimeError):
    """Raised by nothing the caller sees: carried inside ``gather_records_ft``'s return value."""


_GATHER_SEQ = [0]            # collectives are SPMD: every rank numbers its gathers the same way
_DEAD_RANKS: set = set()     # once a rank missed a vote the default group is unusable: stay on the store path
_DEGRADED = [False]          # this rank saw an exchange fail or run in store mode: it says so in every later vote


def rank_timeout_s() -> float:
    """How long a rank that has finished its shard waits for the others before it declares them failed."""
    return float(os.environ.get("VIS_RANK_TIMEOUT_S", "600"))


def _default_store():
    from torch.distributed import distributed_c10d as c10d
    return c10d._get_default_store()


def _all_gather_bytes(dist, payload: bytes) -> List[bytes]:
    """The data-path collective: one length all_gather + one padded-bytes all_gather (RCCL when the group is nccl)."""
    import torch
    backend = dist.get_backend()
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
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
    return [bytes(o[:sz].cpu().numpy().tobytes()) for o, sz in zip(outs, sizes)]


def gather_records_ft(records: List[dict], timeout_s: Optional[float] = None) -> tuple:
    """Fault-tolerant form of the exchange: returns ``(merged records in rank order, sorted list of ranks that did
    not report)``.

    A dead or hung rank must never leave the others blocked inside a collective (a rank failure
    surfaces as failed images, never as a crashed or frozen batch; the reference's per-image try/except at
    src/orchestration/graph.py:349-357 is the single-process form of that rule).  So the collective is preceded by a
    vote through the rendezvous store torch.distributed already runs:
      1. every rank that reaches the exchange sets ``ready/<rank>``;
      2. rank 0 waits for each key up to ``timeout_s`` (VIS_RANK_TIMEOUT_S, default 600) and publishes the list of
         ranks that reported; the others wait for that list;
      3. rank 0 also publishes the MODE of the exchange, and every rank follows the published mode rather than its own
         view: ``collective`` - all ranks reported and none of them has ever seen an exchange fail - moves the payload
         with ONE all_gather pair on the default group (RCCL over xGMI); ``store`` - a rank is missing, or any rank
         reported itself degraded (an earlier exchange raised on it) - lets the survivors exchange their payloads
         through the store (KBs per image); the default group is then not touched again in this process.
    If the coordinator itself is unreachable a rank returns its own records, reports every other rank missing and votes
    ``degraded`` from then on (so the others never wait for it inside a collective)."""
    dist = _dist()
    if dist is None or dist.get_world_size() == 1:
        return list(records), []
    from datetime import timedelta
    W, rank = dist.get_world_size(), dist.get_rank()
    T = rank_timeout_s() if timeout_s is None else float(timeout_s)
    seq = _GATHER_SEQ[0]
    _GATHER_SEQ[0] += 1
    payload = json.dumps(records, default=str).encode("utf-8")
    pre = f"vis/gather{seq}"
    try:
        store = _default_store()
        store.set(f"{pre}/ready/{rank}", b"degraded" if (_DEGRADED[0] or _DEAD_RANKS) else b"ok")
        if rank == 0:
            alive, deadline = [0], time.monotonic() + T
            degraded = bool(_DEGRADED[0] or _DEAD_RANKS)
            for r in range(1, W):
                if r in _DEAD_RANKS:
                    continue
                try:
                    store.wait([f"{pre}/ready/{r}"], timedelta(seconds=max(0.05, deadline - time.monotonic())))
                    alive.append(r)
                    degraded |= bytes(store.get(f"{pre}/ready/{r}")) != b"ok"
                except Exception:
                    logger.error(f"rank {r} did not reach the result exchange within {T:.0f} s: its images are "
                                 f"reported as failed")
            mode = "collective" if (len(alive) == W and not degraded) else "store"
            store.set(f"{pre}/alive", json.dumps({"alive": alive, "mode": mode}).encode())
        else:
            store.wait([f"{pre}/alive"], timedelta(seconds=2 * T + 5))
            verdict = json.loads(bytes(store.get(f"{pre}/alive")).decode())
            alive, mode = verdict["alive"], verdict["mode"]
        dead = sorted(set(range(W)) - set(alive))
        if mode == "collective":
            blobs = _all_gather_bytes(dist, payload)
            if rank == 0 and seq > 0:
                # every rank has voted in THIS exchange, so nobody can still be reading the previous one's keys
                # (an exchange that was not clean leaves the process in store mode for good and never gets here)
                try:
                    for r in range(W):
                        store.delete_key(f"vis/gather{seq - 1}/ready/{r}")
                    store.delete_key(f"vis/gather{seq - 1}/alive")
                except Exception:       # a store without delete_key: the keys are a few bytes per exchange
                    pass
        else:
            _DEGRADED[0] = True
            _DEAD_RANKS.update(dead)
            if rank in alive:
                store.set(f"{pre}/payload/{rank}", payload)
            blobs = []
            for r in range(W):
                if r not in alive:
                    blobs.append(b"")
                    continue
                store.wait([f"{pre}/payload/{r}"], timedelta(seconds=T + 5))
                blobs.append(bytes(store.get(f"{pre}/payload/{r}")))
    except Exception as e:       # coordinator (rank 0 / the store) unreachable: report what this rank has
        logger.error(f"result exchange failed on rank {rank} ({e}); returning this rank's records only", exc_info=True)
        _DEGRADED[0] = True      # published with the next vote: the others then stay out of the collective as well
        return list(records), [r for r in range(W) if r != rank]
    merged: List[dict] = []
    for b in blobs:
        if b:
            merged.extend(json.loads(b.decode("utf-8")))
    return merged, dead


def gather_records(records: List[dict], world: Optional[int] = None) -> List[dict]:
    """All ranks contribute a list of JSON-serialisable records; every rank receives the concatenation in
    rank order (records of ranks that failed to report are simply absent: see ``gather_records_ft``)."""
    return gather_records_ft(records)[0]


def agree_on(value: str, name: str) -> str:
    """Rank 0's ``value`` on every rank, through the rendezvous store (no collective: a rank that died before the
    batch started must not block the others here either)."""
    dist = _dist()
    if dist is None or dist.get_world_size() == 1:
        return value
    from datetime import timedelta
    seq = _GATHER_SEQ[0]
    _GATHER_SEQ[0] += 1
    key = f"vis/agree{seq}/{name}"
    try:
        store = _default_store()
        if dist.get_rank() == 0:
            store.set(key, value.encode())
            return value
        store.wait([key], timedelta(seconds=rank_timeout_s()))
        return bytes(store.get(key)).decode()
    except Exception as e:
        logger.error(f"could not agree on {name} ({e}); keeping the local value")
        return value


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
                            user_notes: Optional[str] = None, prepared: Optional[tuple] = None) -> List[Dict[str, Any]]:
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
    insp_prep, aud_prep = prepared if prepared is not None else (None, None)
    for res, st in zip(get_inspector().analyze_many(image_paths, contexts, prepared=insp_prep), states):
        if res.analysis_failed:
            st["inspector_retry_count"] = 1      # the batched attempt was attempt 1; the node makes the final one
            run_inspector(st)
        else:
            st["current_step"] = "inspector_analysis"
            st["inspector_result"] = res.model_dump()
    for res, st in zip(get_auditor().verify_many(image_paths, contexts, prepared=aud_prep), states):
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
        session_id = agree_on(session_id, "session_id")

    mine: List[dict] = []
    my_idx = list(range(rank, len(image_paths), world))
    pre: Dict[int, Dict[str, Any]] = {}
    if _inspect is run_inspection and len(my_idx) > 1 and _local_batching():
        try:  # shared-decode fast path; any problem falls back to the per-image loop below
            group = int(os.environ.get("VIS_MAX_BATCH", "64"))
            # Request-side encode on the ingest pool (ingest.py), ONE group ahead: while the GPU works on group g the pool
            # encodes group g + 1, and no more than two groups' requests (~1 MB of data URI per image and agent) are ever
            # held in memory - a rank's list may be thousands of images.  The prompt depends on (criticality, domain,
            # notes) only.
            from .agents import get_auditor, get_inspector
            ctx0 = InspectionContext(image_id="prefetch", criticality=criticality, domain=domain, user_notes=user_notes)
            insp, aud = get_inspector(), get_auditor()

            def prepare(g0):
                paths = [image_paths[i] for i in my_idx[g0:g0 + group]]
                return (insp.prepare_many(paths, [ctx0] * len(paths)), aud.prepare_many(paths, [ctx0] * len(paths))) if paths else None

            ahead = prepare(0)
            for g0 in range(0, len(my_idx), group):
                chunk = my_idx[g0:g0 + group]
                cur, ahead = ahead, prepare(g0 + group)
                outs = run_inspections_batched([image_paths[i] for i in chunk], criticality, domain, user_notes, prepared=cur)
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

    records, missing = gather_records_ft(mine)
    for r in missing:      # a rank that died or hung: its images come back as failed, in place (graph.py:349-357)
        for idx in range(r, len(image_paths), world):
            image_path = image_paths[idx]
            image_id = image_id_map[image_path] if image_id_map and image_path in image_id_map \
                else f"{session_id}-{idx:04d}"
            why = f"rank {r} did not report its results within {rank_timeout_s():.0f} s"
            records.append({"image_id": image_id, "image_path": image_path, "error": why, "failure_history": [why],
                            "completed": False, "_index": idx, "_rank": r})
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
