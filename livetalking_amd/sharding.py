"""Session -> GPU sharding.

Sessions are independent units (the reference shares only read-only weights and
avatar banks between them, app.py:62-63,99), so an 8-GPU node runs 8 replicas
of the engine and every session lives on exactly one of them: no collective, no
exchange step, xGMI unused.  Two deployment shapes use the same assignment:

* one process per GPU (bench.py under torch.distributed.run): rank r owns GPU r, its own engine, bank replica and
  `--sessions` sessions (weak scaling: the per-GPU share is fixed, BASELINE.json configs[3] = 16 per GPU);
* one process, several GPUs (`EnginePool`, what the plugin modules' load_model builds): a new session goes to the least
  loaded engine; weights are replicated at load time, avatar banks on first use per engine.
"""
from __future__ import annotations

import threading
from typing import Dict, List, Sequence


class LeastLoaded:
    """Online placement for create_session / remove_session
    (server/session_manager.py:56-94 is the caller-side analogue)."""

    def __init__(self, n_gpus: int, capacity_per_gpu: int = 16):
        self.load = [0] * n_gpus
        self.capacity = capacity_per_gpu
        self.where: Dict[object, int] = {}
        self._lock = threading.Lock()

    def place(self, session_id) -> int:
        with self._lock:
            if session_id in self.where:
                return self.where[session_id]
            g = min(range(len(self.load)), key=lambda i: (self.load[i], i))
            if self.load[g] >= self.capacity:
                raise RuntimeError("all GPUs are at session capacity")   # MaxSessionError analogue
            self.load[g] += 1
            self.where[session_id] = g
            return g

    def release(self, session_id) -> None:
        with self._lock:
            g = self.where.pop(session_id, None)
            if g is not None:
                self.load[g] -= 1


class EnginePool:
    """One engine (weights replica) per GPU inside ONE process - the reference's deployment shape (a single process and
    a single `device`, utils/device.py:4-10, with every session sharing one model object, app.py:62-63,99) widened to the
    node: `engines[i]` lives on `devices[i]`, a session is placed on the least-loaded engine when it is created
    (server/session_manager.py:56-94 is the caller) and released when it goes away.  No data ever crosses engines.
    The same device may be listed more than once (two engines on one GPU: tests)."""

    def __init__(self, devices: Sequence[int], engine_factory, capacity_per_gpu: int = 1 << 30):
        self.devices = list(devices)
        if not self.devices:
            raise ValueError("EnginePool needs at least one device")
        self.engines = [engine_factory(d) for d in self.devices]
        self.placer = LeastLoaded(len(self.devices), capacity_per_gpu)

    def place(self, session_key) -> int:
        """Index of the engine that serves this session (stable for the session's lifetime)."""
        return self.placer.place(session_key)

    def engine_for(self, session_key):
        return self.engines[self.place(session_key)]

    def release(self, session_key) -> None:
        self.placer.release(session_key)

    def load(self) -> List[int]:
        return list(self.placer.load)

    def close(self):
        for e in self.engines:
            e.close()


def visible_devices() -> List[int]:
    """LTK_DEVICES="0,1,2" (explicit list, repeats allowed) > LTK_DEVICE=n (one GPU) > every GPU the process sees."""
    import os
    env = os.environ.get("LTK_DEVICES", "").strip()
    if env:
        return [int(x) for x in env.split(",") if x.strip() != ""]
    if os.environ.get("LTK_DEVICE", "").strip():
        return [int(os.environ["LTK_DEVICE"])]
    try:
        import torch
        n = torch.cuda.device_count()
    except Exception:  # noqa: BLE001
        n = 0
    return list(range(n)) if n > 0 else [0]
