"""Session -> GPU sharding.

Sessions are independent units (the reference shares only read-only weights and
avatar banks between them, app.py:62-63,99), so an 8-GPU node runs 8 replicas
of the engine and every session lives on exactly one of them: no collective, no
exchange step, xGMI unused.  Two deployment shapes use the same assignment:

* one process per GPU (bench.py under torch.distributed.run): rank r serves
  `shard_for_rank(...)`;
* one process, several GPUs (`EnginePool`): `create_session` picks the least
  loaded engine, replicating weights / banks on first use.
"""
from __future__ import annotations

import threading
from typing import Dict, List, Sequence


def assign_round_robin(n_sessions: int, n_gpus: int) -> List[List[int]]:
    """Session ids per GPU; sizes differ by at most one."""
    if n_gpus <= 0:
        raise ValueError("n_gpus must be positive")
    shards: List[List[int]] = [[] for _ in range(n_gpus)]
    for s in range(n_sessions):
        shards[s % n_gpus].append(s)
    return shards


def shard_for_rank(n_sessions: int, world_size: int, rank: int) -> List[int]:
    if not 0 <= rank < world_size:
        raise ValueError("rank outside world")
    return assign_round_robin(n_sessions, world_size)[rank]


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
    """One engine (weights replica) per visible GPU inside one process."""

    def __init__(self, state_dict, devices: Sequence[int], max_frames: int = 256, capacity_per_gpu: int = 16,
                 engine_factory=None):
        if engine_factory is None:
            from .engine import Engine

            def engine_factory(dev):
                e = Engine(dev)
                e.load_wav2lip(state_dict, max_frames=max_frames)
                return e
        self.devices = list(devices)
        self.engines = [engine_factory(d) for d in self.devices]
        self.placer = LeastLoaded(len(self.devices), capacity_per_gpu)

    def engine_for(self, session_id):
        return self.engines[self.placer.place(session_id)]

    def release(self, session_id):
        self.placer.release(session_id)

    def close(self):
        for e in self.engines:
            e.close()
