"""Host-side ingest at batch rates ("move preprocessing to a per-rank worker pool").

The reference handles one image at a time on the calling thread: PIL open -> thumbnail -> JPEG q85 -> base64
(src/agents/vlm_inspector.py:46-88), and the service decodes that JPEG again.  At 20+ images/s per GPU those two
steps - ~20-50 ms of libjpeg / zlib work per 1024x1024 frame and agent - would cost more than the model.  They are
embarrassingly parallel and run inside C code that releases the GIL, so one process-wide thread pool per rank does
them: the request-side encode of EVERY image of a batch is submitted up front (it proceeds while the GPU works on
the first group), and the service-side decode of a group runs while the previous group is in its decode loop.
Threads, not processes: results are large byte strings / arrays (no pickling), and the pool shares the rank's CPU
share (VIS_INGEST_THREADS, default min(4, cpu count): see threads()).
"""
from __future__ import annotations

import itertools
import os
import queue
import threading
import time
from concurrent.futures import Future
from typing import Any, Callable, Iterable, List, Optional, Tuple

# Host timeline of the seam (bench.py's seam64 block): when TRACE is a list, every `span` appends
# (stage, start, end, thread name) - perf_counter seconds.  None (the default) costs one attribute test per stage.
TRACE: Optional[list] = None


class span:
    __slots__ = ("name", "t0")

    def __init__(self, name: str):
        self.name = name

    def __enter__(self):
        self.t0 = time.perf_counter() if TRACE is not None else 0.0
        return self

    def __exit__(self, *exc):
        tr = TRACE
        if tr is not None:
            tr.append((self.name, self.t0, time.perf_counter(), threading.current_thread().name))
        return False


# Priorities (smaller runs first).  The request-side encodes of a whole batch are queued up front; the service-side decode
# of image i must not wait behind the encodes of images i+1.. (the GPU is waiting for it), so it is queued - by `then`,
# only once its encode has finished, never as a task that blocks a thread - with a higher priority.
DECODE, ENCODE = 0, 1


class _Pool:
    """Fixed set of daemon worker threads over a priority queue (ThreadPoolExecutor is FIFO only)."""

    def __init__(self, n: int):
        self.n = n
        self._q: "queue.PriorityQueue" = queue.PriorityQueue()
        self._seq = itertools.count()
        self._stop = False
        self._threads = [threading.Thread(target=self._run, name=f"vis-ingest-{i}", daemon=True) for i in range(n)]
        for t in self._threads:
            t.start()

    def _run(self) -> None:
        while True:
            _, _, item = self._q.get()
            if item is None:
                return
            fut, fn, args, kwargs = item
            if not fut.set_running_or_notify_cancel():
                continue
            try:
                fut.set_result(fn(*args, **kwargs))
            except BaseException as e:      # noqa: BLE001 - delivered through the future
                fut.set_exception(e)

    def submit(self, fn: Callable, *args, priority: int = ENCODE, **kwargs) -> Future:
        fut: Future = Future()
        if self._stop:
            fut.set_exception(RuntimeError("ingest pool was shut down"))
            return fut
        self._q.put((priority, next(self._seq), (fut, fn, args, kwargs)))
        return fut

    def shutdown(self) -> None:
        self._stop = True
        while True:     # pending work is cancelled, running work finishes on its own
            try:
                _, _, item = self._q.get_nowait()
            except queue.Empty:
                break
            if item is not None:
                item[0].cancel()
        for _ in self._threads:
            self._q.put((-1, next(self._seq), None))


_POOL: Optional[_Pool] = None
_LOCK = threading.Lock()


def threads() -> int:
    # default 4: enough for ~27 encodes + decodes per second per rank, and measured faster than 8 or 16 (run_batch_inspection
    # on 64 PNG files: 21.9 / 21.0 / 18.8 / 18.0 images/s with 2 / 4 / 8 / 16 threads) - the threads hold the GIL between their C
    # calls and the one Python thread that issues the kernel launches has to win it back every time
    n = int(os.environ.get("VIS_INGEST_THREADS", "0"))
    return n if n > 0 else max(1, min(4, os.cpu_count() or 1))


def pool() -> _Pool:
    global _POOL
    with _LOCK:
        if _POOL is None:
            _POOL = _Pool(threads())
        return _POOL


def submit(fn: Callable, *args, priority: int = ENCODE, **kwargs) -> Future:
    return pool().submit(fn, *args, priority=priority, **kwargs)


def submit_all(fn: Callable, items: Iterable) -> List[Future]:
    return [submit(fn, *it) if isinstance(it, tuple) else submit(fn, it) for it in items]


def then(first, fn: Callable, priority: int = DECODE) -> Future:
    """Future of ``fn(value)``: queued (with ``priority``) when ``first`` - a Future, or a plain value - is available;
    a failure of ``first`` becomes the failure of the result without ``fn`` running."""
    if not isinstance(first, Future):
        return submit(fn, first, priority=priority)
    out: Future = Future()

    def chain(done: Future) -> None:
        try:
            value = done.result()
        except BaseException as e:      # noqa: BLE001 - also a cancelled encode
            out.set_exception(e)
            return
        def deliver(f: Future) -> None:
            try:
                out.set_result(f.result())
            except BaseException as e:      # noqa: BLE001 - incl. CancelledError after a shutdown
                out.set_exception(e)

        submit(fn, value, priority=priority).add_done_callback(deliver)

    first.add_done_callback(chain)
    return out


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
            _POOL.shutdown()
            _POOL = None
