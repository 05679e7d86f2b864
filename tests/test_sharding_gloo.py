"""Multi-GPU path on the CPU: sessions are independent, so N GPUs = N engine replicas and the only
cross-rank traffic is the bench's barrier + max-over-ranks bookkeeping.

* bench.py's own launcher and the driver's torch.distributed.run launch, world size 2/3 over gloo, through the REAL
  bench code path (`--dry-ranks`: the rank protocol of bench.py with a sleep in place of the GPU step);
* `bench.py --gpus 2` on a box without 2 GPUs fails loudly;
* one process, several engines: the plugin's EnginePool places sessions least-loaded, every session keeps its engine,
  avatar replicas are registered once per (avatar, engine), released sessions free their slot (fake engines)."""
import argparse
import gc
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _json_line(text):
    for line in reversed(text.strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise AssertionError("no JSON line in: " + text[-500:])


def test_bench_spawns_its_own_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "3", "--steps", "4", "--dry-ranks"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 3 and d["ranks_seen"] == 3
    own = d["per_rank_ms_per_step"]                       # rank r sleeps 10 ms x (r + 1) per step
    assert own[0] < own[1] < own[2] and 9.0 < own[0] < 20.0 and 29.0 < own[2] < 45.0
    assert d["ms_per_step"] >= own[2] - 0.5               # the reported time is the MAX over ranks (barrier on both sides)


def test_bench_under_torch_distributed_run():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "3", "--dry-ranks"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["ms_per_step"] >= 19.0


def test_bench_refuses_more_ranks_than_gpus():
    torch = pytest.importorskip("torch")
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(max(have + 1, 2)), "--steps", "1", "--no-also"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "visible GPUs" in r.stderr


def test_engine_pool_places_sessions_across_engines():
    pytest.importorskip("torch")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from fake_engine import FakeEngine
    import livetalking_amd.avatars.wav2lip_avatar as plugin
    from livetalking_amd.sharding import EnginePool
    import synth_inputs as synth

    pool = EnginePool([0, 1, 1], lambda d: FakeEngine(net="tiny", device=d), capacity_per_gpu=2)
    model = plugin.Wav2LipModel(pool)
    avatar = synth.wav2lip_avatar(n_frames=3, full_hw=(120, 160), box=64, seed=0)
    opt = argparse.Namespace(fps=25, batch_size=2, l=10, r=10, sessionid=0)
    sess = [plugin.LipReal(opt, model, avatar) for _ in range(5)]
    assert [s._slot for s in sess] == [0, 1, 2, 0, 1] and pool.load() == [2, 2, 1]
    assert all(s.engine is pool.engines[s._slot] and s.asr.engine is s.engine for s in sess)
    # one bank replica per (avatar, engine), shared by the sessions of that engine
    assert sess[0]._aid == sess[3]._aid and len(pool.engines[0]._avatars) == 1 and len(pool.engines[2]._avatars) == 1
    plugin.LipReal(opt, model, avatar)                     # the 6th fills the pool (capacity 2 per engine)
    gc.collect()                                           # ... and is dropped again at once: its slot is free
    assert pool.load() == [2, 2, 1]
    keep = plugin.LipReal(opt, model, avatar)
    with pytest.raises(RuntimeError):
        plugin.LipReal(opt, model, avatar)
    # a session renders on its own engine only
    feats = np.zeros((2, 80, 16), np.float32)
    import torch
    before = [e.calls["wav2lip_infer"] for e in pool.engines]
    sess[2].inference_batch(0, torch.from_numpy(feats))
    after = [e.calls["wav2lip_infer"] for e in pool.engines]
    assert [a - b for a, b in zip(after, before)] == [0, 0, 1]
    del sess[0]
    gc.collect()
    assert pool.load()[0] == 1 and keep._slot == 2         # removed sessions free their slot (weakref.finalize)


def test_bench_session_threads_run_free_between_steps():
    """bench.py's SessionThreads: with several sessions every thread issues its `nsteps` calls back to back (no hand-shake
    between steps - the reference's per-session inference threads, base_avatar.py:326-381); indices advance per step and
    per session; an exception in one session surfaces in the caller."""
    import threading
    import time
    import bench

    class Sess:
        batch_size = 4

        def __init__(self, delay):
            self.calls, self.delay = [], delay

        def inference_batch(self, index, feats):
            self.calls.append((index, time.perf_counter()))
            time.sleep(self.delay)
            return [None] * self.batch_size

    slow, fast = Sess(0.02), Sess(0.001)
    drv = bench.SessionThreads([slow, fast], [None, None], stride=7)
    try:
        drv.step(0)                                   # one lock-step call each (warm-up style)
        t0 = time.perf_counter()
        drv.step(1, nsteps=5)
        dt = time.perf_counter() - t0
    finally:
        drv.close()
    assert [c[0] for c in slow.calls] == [0 * 4 + 0, 4, 8, 12, 16, 20]
    assert [c[0] for c in fast.calls] == [7, 11, 15, 19, 23, 27]
    # the fast session finished its five calls while the slow one was still in its first or second: no per-step barrier
    assert fast.calls[-1][1] < slow.calls[3][1]
    assert dt < 5 * 0.02 + 0.5           # (generous: the build container's 8 cores are shared; the ordering assert above is the property)
    assert drv.frames == [24, 24]

    class Bad(Sess):
        def inference_batch(self, index, feats):
            raise RuntimeError("boom")

    drv = bench.SessionThreads([Sess(0.0), Bad(0.0)], [None, None], stride=1)
    try:
        with pytest.raises(RuntimeError):
            drv.step(0, nsteps=2)
    finally:
        drv.close()


def test_rank_device_override_is_explicit(monkeypatch):
    """LTK_RANK_DEVICES maps local ranks onto listed GPUs (the one-GPU test of the real multi-rank path); without it rank r is
    GPU r, and a rank the list does not name is an error, not GPU 0."""
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.delenv("LTK_RANK_DEVICES", raising=False)
    assert [bench.rank_device(r) for r in range(4)] == [0, 1, 2, 3]
    monkeypatch.setenv("LTK_RANK_DEVICES", "0,0")
    assert [bench.rank_device(r) for r in range(2)] == [0, 0]
    with pytest.raises(SystemExit):
        bench.rank_device(2)


@pytest.mark.gpu
def test_bench_two_real_ranks_on_one_gpu():
    """The NON-dry multi-rank path of bench.py - engine + bank replica + session per rank, gloo barriers on both sides of a GPU
    step, max over ranks - under the driver's own launcher (torch.distributed.run), with both ranks mapped onto GPU 0 by the
    explicit LTK_RANK_DEVICES override: n_gpus 2, two per-rank rates, value = 2 x frames / max time, i.e. about the sum of
    the two ranks' rates (they share one GPU here, so each runs at roughly half the single-rank rate)."""
    import socket
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, LTK_RANK_DEVICES="0,0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "3", "--warmup", "2", "--no-also", "--no-cpu-baseline", "--no-traffic",
           "--sustain", "0.3"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    assert d["metric"] == "inferfps" and d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak"
    assert len(d["per_rank_fps"]) == 2 and all(v > 100 for v in d["per_rank_fps"])
    assert d["config"]["parallelism"].startswith("session-sharded x2")
    frames = 2 * 3 * 16
    assert abs(d["value"] - frames / (d["ms_per_step"] * 3e-3)) <= 0.01 * d["value"]           # whole-job frames / max-over-ranks time
    assert d["value"] <= sum(d["per_rank_fps"]) * 1.02                                         # never more than the ranks' own rates add up to
    assert d["value"] >= 0.5 * sum(d["per_rank_fps"])                                          # and the barrier costs less than half of it
    assert d["sustained"]["steps"] >= 3 and d["sustained"]["value"] > 100
    # without the override two ranks on a one-GPU box must fail loudly
    if torch.cuda.device_count() < 2:
        env2 = {k: v for k, v in env.items() if k != "LTK_RANK_DEVICES"}
        r2 = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env2)
        assert r2.returncode != 0 and "needs GPU 1" in (r2.stderr + r2.stdout)
