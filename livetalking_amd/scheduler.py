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
* LTK_COALESCE_MS > 0 holds a batch open for that long after its first request (fixed window) when that request finds the
  engine idle (it leads the batch in its own thread); requests that arrive while a call is in flight are issued by the workers
  on the lead timer below, without the window;
* up to LTK_INFLIGHT (default 2) calls are in flight: the next batch is issued shortly before the running call is expected to
  end, so its launches queue behind the running kernels and the GPU does not idle through the host turn-around.

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
    """Continuous batching with up to LTK_INFLIGHT (default 2) engine calls in flight.

    The engine serialises calls on one stream and releases its enqueue lock before it waits (csrc/engine.hip,
    ltk_wav2lip_infer), so a second call issued while the first still runs queues its launches right behind the first
    call's kernels: the GPU does not idle through the host turn-around between two calls (completion wake-up, Python,
    ctypes, launch: ~75 us of a 1.3-ms step, ~1 ms of a 16-ms 256-frame step with 16 session threads).  To keep batches
    as large as they are with one call in flight, the second call is not issued when the first request arrives but
    LTK_INFLIGHT_LEAD_US (default 300) before the running call is EXPECTED to end - from a running seconds-per-frame
    estimate of the calls' device time - and takes everything queued by then."""

    def __init__(self, engine, kind: str = "wav2lip", window_ms: float = 0.0, max_frames: int = 0):
        self.engine = engine
        self.kind = kind
        self._call = engine.wav2lip_infer if kind == "wav2lip" else engine.musetalk_infer
        self.window = max(0.0, float(window_ms)) * 1e-3
        self._max_frames = int(max_frames)
        self._cv = threading.Condition()
        self._pending = collections.deque()
        self._inflight = 0                 # engine calls issued and not yet returned (leaders + workers)
        self._inflight_frames = 0          # frames of those calls
        self._lead_frac = max(0.0, float(os.environ.get("LTK_INFLIGHT_LEAD_FRAC", "0.1")))
        self._max_inflight = max(1, int(os.environ.get("LTK_INFLIGHT", "2")))
        self._lead = max(0.0, float(os.environ.get("LTK_INFLIGHT_LEAD_US", "300"))) * 1e-6
        self._spf = None                   # seconds per frame of a call that had the engine to itself (EWMA)
        self._busy_until = 0.0             # perf_counter() at which the calls in flight are expected to be done
        self._last_done = 0.0              # perf_counter() of the latest completion (calls complete in issue order: one stream)
        self._leader_hold = False          # a leader is holding its batch open (window): workers leave the queue alone
        self._closed = False
        self._workers = []
        self._idle_exit = max(0.05, float(os.environ.get("LTK_WORKER_IDLE_S", "5")))
        self._auto_window = max(0.0, float(os.environ.get("LTK_COALESCE_AUTO_US", "200"))) * 1e-6
        self._last_multi = -1e9            # perf_counter() of the last batch that carried more than one request
        self.stats = {"calls": 0, "requests": 0, "frames": 0, "max_requests_per_call": 0, "overlapped_calls": 0}

    def _limit(self) -> int:
        if self._max_frames > 0:
            return self._max_frames
        lim = getattr(self.engine, "max_frames" if self.kind == "wav2lip" else "mt_max_frames", 0)
        return int(lim) if lim else 1 << 30

    def infer(self, aid, index, batch, in_ptr, out_ptr):
        """Blocks until this request's frames are ready.  Raises what the engine raised for the call that carried it."""
        batch = int(batch)
        with self._cv:
            # A lone session (nothing in flight, nothing queued, no window to hold, no multi-request batch lately): the call goes straight
            # down in this thread with no request object, queue or event - the Python between two calls of a single session is time
            # the GPU idles (scripts/host_overhead.py: 15 -> 5 us of the scheduler's share).
            solo = (self._inflight == 0 and not self._pending and self.window <= 0.0 and
                    (self._auto_window <= 0.0 or time.perf_counter() - self._last_multi >= 2.0))
            if solo:
                self._inflight += 1
                self._inflight_frames += batch
                if self._spf is not None:
                    self._busy_until = time.perf_counter() + batch * self._spf
        if solo:
            return self._run_solo((aid, index, batch, in_ptr, out_ptr), batch)
        r = _Req((aid, index, batch, in_ptr, out_ptr), batch)
        with self._cv:
            self._pending.append(r)
            leader = self._inflight == 0 and len(self._pending) == 1
            if leader:
                self._inflight += 1
                self._leader_hold = True
            else:
                self._ensure_workers()
                self._cv.notify_all()
        if leader:
            # idle engine: run the call in this thread (no hop); whatever queues up meanwhile is the workers' business
            if self.window > 0.0:
                time.sleep(self.window)
            elif self._auto_window > 0.0 and time.perf_counter() - self._last_multi < 2.0:
                time.sleep(self._auto_window)          # other sessions are active: let the ones due now join this launch
            self._run_one_batch(alone=True)
        r.done.wait()
        if r.err is not None:
            raise r.err

    def _run_solo(self, args, frames):
        """One request, one call, in the caller's thread; holds the in-flight slot infer() took.  Sessions that arrive meanwhile queue
        up as usual (they see a call in flight) and the workers issue their batch - behind this call, or when it ends."""
        t0 = time.perf_counter()
        err = None
        try:
            self._call([args])
        except Exception as ex:  # noqa: BLE001
            err = ex
        t1 = time.perf_counter()
        with self._cv:
            self._inflight -= 1
            st = self.stats
            st["calls"] += 1
            st["requests"] += 1
            st["frames"] += frames
            if st["max_requests_per_call"] < 1:
                st["max_requests_per_call"] = 1
            if err is None and frames > 0:
                spf = (t1 - max(t0, self._last_done)) / frames
                self._spf = spf if self._spf is None else 0.75 * self._spf + 0.25 * spf
            self._last_done = t1
            self._inflight_frames -= frames
            self._busy_until = t1 + (self._inflight_frames * self._spf if self._spf is not None else 0.0)
            if self._pending:
                self._ensure_workers()
                self._cv.notify_all()
        if err is not None:
            raise err

    def _ensure_workers(self):
        """Under the lock."""
        self._workers = [w for w in self._workers if w.is_alive()]
        while len(self._workers) < self._max_inflight:
            w = threading.Thread(target=self._work, name="ltk-batch", daemon=True)
            self._workers.append(w)
            w.start()

    def _take_batch(self):
        """Under the lock: the queued requests that fit one engine call (FIFO)."""
        group, frames, lim = [], 0, self._limit()
        while self._pending and (not group or frames + self._pending[0].batch <= lim):
            q = self._pending.popleft()
            group.append(q)
            frames += q.batch
        return group, frames

    def _run_one_batch(self, alone: bool):
        """The caller holds one in-flight slot (self._inflight already counts it); released here."""
        with self._cv:
            if alone:
                self._leader_hold = False
            group, frames = self._take_batch()
            if alone and self._pending:
                # the leader's batch hit the frame limit: the rest is the workers' business from NOW on (they skipped the queue
                # while the batch was held open), not only once this call completes
                self._ensure_workers()
                self._cv.notify_all()
            now = time.perf_counter()
            self._inflight_frames += frames
            if self._spf is not None:
                self._busy_until = max(now, self._busy_until) + frames * self._spf
            if not alone:
                self.stats["overlapped_calls"] += 1
        t0 = time.perf_counter()
        errs = [None] * len(group)
        if group:
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
        t1 = time.perf_counter()
        with self._cv:
            self._inflight -= 1
            if group:
                self.stats["calls"] += 1
                self.stats["requests"] += len(group)
                self.stats["frames"] += frames
                self.stats["max_requests_per_call"] = max(self.stats["max_requests_per_call"], len(group))
                if len(group) > 1:
                    self._last_multi = t1
                if frames > 0 and all(e is None for e in errs):
                    # the engine runs calls in issue order: this one had the GPU from its issue or from the previous completion on
                    spf = (t1 - max(t0, self._last_done)) / frames
                    self._spf = spf if self._spf is None else 0.75 * self._spf + 0.25 * spf
            self._last_done = t1
            self._inflight_frames -= frames
            # re-anchor the prediction at every completion (calls complete in issue order): what is still in flight starts now
            self._busy_until = t1 + (self._inflight_frames * self._spf if self._spf is not None else 0.0)
            if self._pending:
                self._ensure_workers()
            self._cv.notify_all()
        for q, err in zip(group, errs):
            q.err = err
            q.done.set()

    def _work(self):
        """Worker: issues a call whenever requests are queued and either nothing is in flight, or a slot is free and the calls in
        flight are about to end.  Exits after LTK_WORKER_IDLE_S idle seconds (a later request starts a new one), so an unused
        scheduler does not pin its engine for the life of the process."""
        me = threading.current_thread()
        while True:
            with self._cv:
                idle_since = time.perf_counter()
                while True:
                    if self._closed:
                        if me in self._workers:
                            self._workers.remove(me)
                        return
                    now = time.perf_counter()
                    if self._pending and not self._leader_hold:
                        if self._inflight == 0:
                            break
                        if self._inflight < self._max_inflight and self._spf is not None:
                            # lead: the fixed floor, or a tenth of what is in flight (thread wake-ups under 16 session threads are not
                            # 300-us precise; the requests of the sessions that are not in flight have all arrived long before)
                            lead = max(self._lead, self._lead_frac * self._inflight_frames * self._spf)
                            wait_t = self._busy_until - lead - now
                            if wait_t <= 0.0:
                                break
                            self._cv.wait(timeout=wait_t)
                            continue
                        self._cv.wait()                 # a completion (or close) wakes us
                        idle_since = time.perf_counter()
                        continue
                    if now - idle_since >= self._idle_exit:
                        if me in self._workers:
                            self._workers.remove(me)
                        return
                    self._cv.wait(timeout=self._idle_exit)
                alone = self._inflight == 0
                self._inflight += 1
            self._run_one_batch(alone=alone)

    def close(self):
        """Stop the workers.  Requests still queued are failed (their callers raise) instead of being left blocked; a later
        infer() still works: it runs as a leader and starts new workers when it needs them."""
        with self._cv:
            self._closed = True
            dropped = list(self._pending)
            self._pending.clear()
            workers = list(self._workers)
            self._cv.notify_all()
        for q in dropped:
            q.err = RuntimeError("scheduler closed")
            q.done.set()
        for w in workers:
            if w is not threading.current_thread():
                w.join(timeout=5.0)
        with self._cv:
            self._workers = [w for w in self._workers if w.is_alive()]
            self._closed = False
            if self._pending:
                # a request that arrived while the workers were shutting down (the ones started for it saw `_closed` and
                # left): it is served by fresh workers now instead of waiting for some later infer() to come by
                self._ensure_workers()
                self._cv.notify_all()


class CoalescingScheduler(BatchingScheduler):
    def __init__(self, engine, window_ms: float, kind: str = "wav2lip"):
        super().__init__(engine, kind, window_ms)


# one scheduler per (engine, kind), held on the engine object itself (a registry keyed by id(engine) kept every engine - and its
# GPU weights - alive for the life of the process).  Engine.close() closes them; their worker threads also exit when idle.
_LOCK = threading.Lock()


def get_scheduler(engine, kind: str = "wav2lip"):
    with _LOCK:
        table = engine.__dict__.setdefault("_ltk_schedulers", {})
        s = table.get(kind)
        if s is None:
            s = BatchingScheduler(engine, kind, float(os.environ.get("LTK_COALESCE_MS", "0")))
            table[kind] = s
        return s
