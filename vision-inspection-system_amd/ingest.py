"""Host-side ingest at batch rates ("move preprocessing to a per-rank worker pool").

The reference handles one image at a time on the calling thread: PIL open -> thumbnail -> JPEG q85 -> base64
(src/agents/vlm_inspector.py:46-88), and the service decodes that JPEG again.  At 20+ images/s per GPU those two
steps - ~20-50 ms of libjpeg / zlib work per 1024x1024 frame and agent - would cost more than the model.  They are
embarrassingly parallel and run inside C code that releases the GIL, so one process-wide thread pool per rank does
them: the request-side encode of EVERY image of a batch is submitted up front (it proceeds while the GPU works on
the first group), and the service-side decode of a group runs while the previous group is in its decode loop.
Threads, not processes: results are large byte strings / arrays (no pickling), and the pool shares the rank's CPU
share (VIS_INGEST_THREADS, default min(16, cpu count)).
"""
from __future__ import annotations

import os
import threading
from concurrent.futures import Future, ThreadPoolExecutor
from typing import Any, Callable, Iterable, List, Optional, Tuple

_POOL: Optional[ThreadPoolExecutor] = None
_LOCK = threading.Lock()


def threads() -> int:
    n = int(os.environ.get("VIS_INGEST_THREADS", "0"))
    return n if n > 0 else max(1, min(16, os.cpu_count() or 1))


def pool() -> ThreadPoolExecutor:
    global _POOL
    with _LOCK:
        if _POOL is None:
            _POOL = ThreadPoolExecutor(max_workers=threads(), thread_name_prefix="vis-ingest")
        return _POOL


def submit(fn: Callable, *args, **kwargs) -> Future:
    return pool().submit(fn, *args, **kwargs)


def submit_all(fn: Callable, items: Iterable) -> List[Future]:
    p = pool()
    return [p.submit(fn, *it) if isinstance(it, tuple) else p.submit(fn, it) for it in items]


def outcome(fut: Future) -> Tuple[bool, Any]:
    """(True, result) or (False, exception): a failed item stays a per-item result (the agents never raise)."""
    try:
        return True, fut.result()
    except Exception as e:       # noqa: BLE001 - the caller turns it into an analysis_failed result
        return False, e


def shutdown() -> None:
    global _POOL
    with _LOCK:
        if _POOL is not None:
            _POOL.shutdown(wait=False, cancel_futures=True)
            _POOL = None
