"""Cross-session batching of `inference_batch` calls.

N session threads call `infer()` concurrently (avatars/base_avatar.py:366: one inference thread per session, all
sharing one model object, app.py:62-63,99).  The engine runs one call at a time per GPU, so requests that arrive while
a call is in flight would otherwise queue behind it one by one, each as its own 16-frame launch sequence.  The
scheduler batches them instead (continuous batching):

* an idle engine takes a request immediately, in the caller's thread - a single session pays no hop and no window;
* while a call is in flight, new requests collect in a queue; when the call returns, everything queued (up to the
  engine's max_frames) goes down as ONE engine call with nreq > 1, which is how 16 sessions per GPU fill the chip
  (256 frames per launch sequence instead of 16);
* while several sessions are active (a batch of more than one request formed within the last two seconds) a leader holds its
  batch open for LTK_COALESCE_AUTO_US (default 200 us, 0.03 % of a 640-ms step; 0 / 200 / 500 us measured with 16 free-running
  session threads: 200 >= 0 > 500, profiles/r02_scheduler_window.txt) so that sessions woken by the same clock
  tick ride one launch sequence instead of "one alone, then the rest"; a lone session never waits;
* LTK_COALESCE_MS > 0 holds every batch open for that long after its first request (fixed window).

A session's frames keep their order: a request is one contiguous (index .. index+batch) span, a session has at most
one request in flight (its inference thread blocks in `infer()`), and every request's frames land in its own output
tensor.  One scheduler per (engine, kind); kinds: "wav2lip" -> Engine.wav2lip_infer, "musetalk" -> Engine.musetalk_infer.
"""
from __future__ import annotations

import collections
import os
import threading
import time


class _Req:
    __slots__ = ("args", "batch", "done", "err")

    def __init__(self, args, batch):
        self.args, self.batch = args, batch
        self.done = threading.Event()
        self.err = None


class BatchingScheduler:
    def __init__(self, engine, kind: str = "wav2lip", window_ms: float = 0.0, max_frames: int = 0):
        self.engine = engine
        self.kind = kind
        self._call = engine.wav2lip_infer if kind == "wav2lip" else engine.musetalk_infer
        self.window = max(0.0, float(window_ms)) * 1e-3
        self._max_frames = int(max_frames)
        self._cv = threading.Condition()
        self._pending = collections.deque()
        self._busy = False
        self._handoff = False
        self._closed = False
        self._worker = None
        self._auto_window = max(0.0, float(os.environ.get("LTK_COALESCE_AUTO_US", "200"))) * 1e-6
        self._last_multi = -1e9            # perf_counter() of the last batch that carried more than one request
        self.stats = {"calls": 0, "requests": 0, "frames": 0, "max_requests_per_call": 0}

    def _limit(self) -> int:
        if self._max_frames > 0:
            return self._max_frames
        lim = getattr(self.engine, "max_frames" if self.kind == "wav2lip" else "mt_max_frames", 0)
        return int(lim) if lim else 1 << 30

    def infer(self, aid, index, batch, in_ptr, out_ptr):
        """Blocks until this request's frames are ready.  Raises what the engine raised for the call that carried it."""
        r = _Req((aid, index, batch, in_ptr, out_ptr), int(batch))
        with self._cv:
            self._pending.append(r)
            leader = not self._busy
            if leader:
                self._busy = True
        if leader:
            # idle engine: run the call in this thread (no hop); whatever queued up meanwhile goes to the worker
            if self.window > 0.0:
                time.sleep(self.window)
            elif self._auto_window > 0.0 and time.perf_counter() - self._last_multi < 2.0:
                time.sleep(self._auto_window)          # other sessions are active: let the ones due now join this launch
            self._run_one_batch()
            with self._cv:
                if self._pending:
                    self._handoff = True
                    if self._worker is None:
                        self._worker = threading.Thread(target=self._work, name="ltk-batch", daemon=True)
                        self._worker.start()
                    self._cv.notify_all()
                else:
                    self._busy = False
        r.done.wait()
        if r.err is not None:
            raise r.err

    def _take_batch(self):
        """Under the lock: the queued requests that fit one engine call (FIFO)."""
        group, frames, lim = [], 0, self._limit()
        while self._pending and (not group or frames + self._pending[0].batch <= lim):
            q = self._pending.popleft()
            group.append(q)
            frames += q.batch
        return group, frames

    def _run_one_batch(self):
        with self._cv:
            group, frames = self._take_batch()
        if not group:
            return
        errs = [None] * len(group)
        try:
            self._call([q.args for q in group])
        except Exception as ex:  # noqa: BLE001
            if len(group) == 1:
                errs[0] = ex
            else:
                # One bad request (released avatar, bad pointer, oversize batch) must not kill the inference threads of the
                # sessions it happened to be batched with (base_avatar.py:366 has no try/except): the engine validates a
                # call before it launches anything, so the requests are re-issued one by one and only the offender raises.
                for i, q in enumerate(group):
                    try:
                        self._call([q.args])
                    except Exception as ex_i:  # noqa: BLE001
                        errs[i] = ex_i
        with self._cv:
            self.stats["calls"] += 1
            self.stats["requests"] += len(group)
            self.stats["frames"] += frames
            self.stats["max_requests_per_call"] = max(self.stats["max_requests_per_call"], len(group))
            if len(group) > 1:
                self._last_multi = time.perf_counter()
        for q, err in zip(group, errs):
            q.err = err
            q.done.set()

    def _work(self):
        """Worker: owns the engine between a leader's hand-off and the moment the queue runs empty."""
        while True:
            with self._cv:
                while not self._handoff and not self._closed:
                    self._cv.wait()
                if self._closed and not self._handoff:
                    self._worker = None
                    return
                self._handoff = False
            while True:
                with self._cv:
                    if not self._pending:
                        self._busy = False
                        break
                self._run_one_batch()

    def close(self):
        """Stop the worker.  Requests already queued are failed (their callers raise) instead of being left blocked; a later
        infer() still works: it runs as a leader and starts a new worker when it needs one."""
        with self._cv:
            self._closed = True
            dropped = []
            if not self._busy or self._handoff:           # nobody is about to drain the queue
                dropped = list(self._pending)
                self._pending.clear()
                if self._handoff:
                    self._handoff = False
                    self._busy = False
            self._cv.notify_all()
        for q in dropped:
            q.err = RuntimeError("scheduler closed")
            q.done.set()
        w = self._worker
        if w is not None and w is not threading.current_thread():
            w.join(timeout=5.0)
        with self._cv:
            if self._worker is not None and not self._worker.is_alive():
                self._worker = None
            self._closed = False


class CoalescingScheduler(BatchingScheduler):
    def __init__(self, engine, window_ms: float, kind: str = "wav2lip"):
        super().__init__(engine, kind, window_ms)


# one scheduler per (engine, kind), held on the engine object itself: it goes away with the engine (a registry keyed by
# id(engine) kept every engine - and its GPU weights - alive for the life of the process)
_LOCK = threading.Lock()


def get_scheduler(engine, kind: str = "wav2lip"):
    with _LOCK:
        table = engine.__dict__.setdefault("_ltk_schedulers", {})
        s = table.get(kind)
        if s is None:
            s = BatchingScheduler(engine, kind, float(os.environ.get("LTK_COALESCE_MS", "0")))
            table[kind] = s
        return s
