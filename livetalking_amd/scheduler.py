"""Cross-session batching of `inference_batch` calls.

N session threads call `infer()` concurrently (avatars/base_avatar.py:366, one
inference thread per session).  With LTK_COALESCE_MS unset (default 0) every
call goes straight to the engine - the engine already lets calls from
different threads queue back-to-back on its compute stream.  With
LTK_COALESCE_MS > 0 a dispatcher thread gathers the requests that arrive within
that window (bounded by the engine's max_frames) and issues ONE
ltk_wav2lip_infer with nreq > 1, so small per-session batches fill the GPU.
A session's frames keep their order: a request is one contiguous
(index .. index+batch) span and each session has at most one request in flight.
"""
from __future__ import annotations

import os
import queue
import threading
import time


class _Req:
    __slots__ = ("aid", "index", "batch", "mel", "out", "done", "err")

    def __init__(self, aid, index, batch, mel, out):
        self.aid, self.index, self.batch, self.mel, self.out = aid, index, batch, mel, out
        self.done = threading.Event()
        self.err = None


class DirectScheduler:
    def __init__(self, engine):
        self.engine = engine

    def infer(self, aid, index, batch, mel_ptr, out_ptr):
        self.engine.wav2lip_infer([(aid, index, batch, mel_ptr, out_ptr)])

    def close(self):
        pass


class CoalescingScheduler:
    def __init__(self, engine, window_ms: float):
        self.engine = engine
        self.window = window_ms * 1e-3
        self.q: "queue.Queue[_Req]" = queue.Queue()
        self._stop = False
        self._thread = threading.Thread(target=self._run, name="ltk-coalesce", daemon=True)
        self._thread.start()

    def infer(self, aid, index, batch, mel_ptr, out_ptr):
        r = _Req(aid, index, batch, mel_ptr, out_ptr)
        self.q.put(r)
        r.done.wait()
        if r.err is not None:
            raise r.err

    def _run(self):
        while not self._stop:
            try:
                first = self.q.get(timeout=0.1)
            except queue.Empty:
                continue
            group, frames = [first], first.batch
            deadline = time.perf_counter() + self.window
            while frames < self.engine.max_frames:
                left = deadline - time.perf_counter()
                if left <= 0:
                    break
                try:
                    r = self.q.get(timeout=left)
                except queue.Empty:
                    break
                if frames + r.batch > self.engine.max_frames:
                    self.q.put(r)
                    break
                group.append(r)
                frames += r.batch
            try:  # every request's frames land in its own output tensor (ltk_w2l_req.d_pred)
                self.engine.wav2lip_infer([(r.aid, r.index, r.batch, r.mel, r.out) for r in group])
            except Exception as ex:  # noqa: BLE001 - hand the error to every waiting caller
                for r in group:
                    r.err = ex
            for r in group:
                r.done.set()

    def close(self):
        self._stop = True


_SCHEDULERS = {}
_LOCK = threading.Lock()


def get_scheduler(engine):
    with _LOCK:
        s = _SCHEDULERS.get(id(engine))
        if s is None:
            ms = float(os.environ.get("LTK_COALESCE_MS", "0"))
            s = CoalescingScheduler(engine, ms) if ms > 0 else DirectScheduler(engine)
            _SCHEDULERS[id(engine)] = s
        return s
