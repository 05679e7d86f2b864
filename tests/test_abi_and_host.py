"""CPU: the C-ABI library loads and exports every symbol include/ltk.h declares
(no compute calls - there is no GPU here), and the host-side mirror of the
reference's plugin interface behaves like the reference's."""
import argparse
import ctypes
import os
import re
import sys
import threading
import time
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "ltk.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(ltk_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from livetalking_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    declared = _declared_symbols()
    assert len(declared) >= 14
    assert sorted(_lib.SYMBOLS) == declared, "ctypes binding and include/ltk.h disagree"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} not exported"
    loaded = _lib.load()
    assert loaded.ltk_version().startswith(b"ltk_hip")


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from livetalking_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_window_starts_match_reference_slicing():
    from livetalking_amd.avatars.audio_features.mel import window_starts
    from oracle import mel_oracle
    assert window_starts(52, 10, 10, 25, 84) == mel_oracle.window_starts(52, 10, 10)
    assert window_starts(22, 10, 10, 25, 36) == [16]
    # tail clamp (mel.py:58-59): a window that would overrun starts at n_cols-16
    assert window_starts(52, 10, 10, 25, 70)[-1] == 54


class FakeEngine:
    """Stands in for the HIP engine: records calls, fills nothing."""
    device = 0
    max_frames = 64
    torch_device = "cpu"

    def __init__(self):
        self.mel_calls, self.infer_calls = [], []
        self.lock = threading.Lock()

    def mel_step(self, pcm, starts, ptr, stream=0):
        self.mel_calls.append((np.array(pcm, copy=True), list(starts)))

    def wav2lip_infer(self, reqs, stream=0):
        with self.lock:
            self.infer_calls.append(list(reqs))


def test_melasr_protocol_matches_reference_cadence(monkeypatch):
    """Same queue traffic as mel.py:34-67: 2B chunks out per step, one feature batch
    once context exists, l+r chunks retained."""
    torch = pytest.importorskip("torch")
    from livetalking_amd.avatars.audio_features import mel as melmod
    opt = argparse.Namespace(fps=25, batch_size=16, l=10, r=10)
    eng = FakeEngine()
    asr = melmod.MelASR(opt, None, engine=eng)
    monkeypatch.setattr(asr, "_torch", types.SimpleNamespace(
        empty=lambda shape, dtype=None, device=None: types.SimpleNamespace(shape=shape, data_ptr=lambda: 0),
        float32=None, device=lambda *a: None))
    rng = np.random.default_rng(0)
    chunks = [rng.standard_normal(320).astype(np.float32) for _ in range(20 + 64)]
    for c in chunks:
        asr.put_audio_frame(c, {})
    asr.warm_up()
    assert asr.output_queue.qsize() == 10 and len(asr.frames) == 20
    asr.run_step()
    assert asr.output_queue.qsize() == 10 + 32 and asr.feat_queue.qsize() == 1 and len(asr.frames) == 20
    pcm, starts = eng.mel_calls[0]
    assert pcm.shape == (16640,) and np.array_equal(pcm, np.concatenate(chunks[:52]))
    assert starts == [16, 19, 22, 25, 28, 32, 35, 38, 41, 44, 48, 51, 54, 57, 60, 64]
    asr.run_step()
    assert np.array_equal(eng.mel_calls[1][0], np.concatenate(chunks[32:84]))
    # silence synthesis when the queue runs dry: zeros, type 1 (base_asr.py:66-69)
    f = asr.get_audio_frame()
    assert f.type == 1 and not f.data.any()


def test_coalescing_scheduler_groups_concurrent_sessions():
    pytest.importorskip("torch")
    from livetalking_amd import scheduler
    eng = FakeEngine()
    sch = scheduler.CoalescingScheduler(eng, window_ms=200.0)
    threads = [threading.Thread(target=sch.infer, args=(1, 16 * i, 16, 1000 + i, 2000 + i)) for i in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=10)
    sch.close()
    total = sorted(r for call in eng.infer_calls for r in call)
    assert total == [(1, 0, 16, 1000, 2000), (1, 16, 16, 1001, 2001), (1, 32, 16, 1002, 2002)]
    assert len(eng.infer_calls) < 3, "requests inside the window must share a launch"


def test_continuous_batching_without_a_window():
    """Default scheduler (no window): an idle engine takes a request at once in the caller's thread; requests that arrive
    while that call is in flight go down together as ONE call afterwards; errors reach exactly the callers of the failed call."""
    pytest.importorskip("torch")
    from livetalking_amd import scheduler

    class SlowEngine(FakeEngine):
        def wav2lip_infer(self, reqs, stream=0):
            if any(r[0] == 99 for r in reqs):
                raise RuntimeError("boom")
            time.sleep(0.15)
            super().wav2lip_infer(reqs, stream)

    eng = SlowEngine()
    sch = scheduler.BatchingScheduler(eng, "wav2lip")
    first = threading.Thread(target=sch.infer, args=(1, 0, 16, 1000, 2000))
    first.start()
    time.sleep(0.05)                                     # the first call is now in flight
    late = [threading.Thread(target=sch.infer, args=(1, 16 * i, 16, 1000 + i, 2000 + i)) for i in (1, 2, 3)]
    for t in late:
        t.start()
    for t in [first] + late:
        t.join(timeout=10)
        assert not t.is_alive()
    assert [len(c) for c in eng.infer_calls] == [1, 3], eng.infer_calls
    assert sorted(r[1] for r in eng.infer_calls[1]) == [16, 32, 48]
    assert sch.stats["max_requests_per_call"] == 3 and sch.stats["frames"] == 64
    # a lone request right after: again alone, again in the caller's thread
    sch.infer(1, 64, 16, 1, 2)
    assert len(eng.infer_calls) == 3 and len(eng.infer_calls[2]) == 1
    with pytest.raises(RuntimeError, match="boom"):
        sch.infer(99, 0, 16, 1, 2)
    sch.infer(1, 80, 16, 1, 2)                           # the scheduler survives a failed call
    sch.close()


def test_two_calls_in_flight_keep_order_and_batch_size(monkeypatch):
    """Up to LTK_INFLIGHT = 2 engine calls in flight (the engine serialises them on one stream and waits outside its enqueue lock):
    the next batch goes down shortly BEFORE the running call is expected to end, so it still carries everything that queued up
    meanwhile.  Eight free-running session threads against an engine that needs 1 ms per frame and serialises like the real one:
    every request is served exactly once and in its session's order, never more than two calls are inside the engine, calls do
    overlap, and batches stay multi-request.  LTK_INFLIGHT=1 restores one call at a time."""
    pytest.importorskip("torch")
    from livetalking_amd import scheduler

    class SerialEngine(FakeEngine):
        def __init__(self):
            super().__init__()
            self.gpu = threading.Lock()          # "the stream": one call's device work at a time, in issue order
            self.inside = 0
            self.peak = 0

        def wav2lip_infer(self, reqs, stream=0):
            with self.lock:
                self.inside += 1
                self.peak = max(self.peak, self.inside)
                self.infer_calls.append(list(reqs))
            with self.gpu:
                time.sleep(1e-3 * sum(r[2] for r in reqs))
            with self.lock:
                self.inside -= 1

    def run(inflight):
        monkeypatch.setenv("LTK_INFLIGHT", str(inflight))
        eng = SerialEngine()
        sch = scheduler.BatchingScheduler(eng, "wav2lip")
        S, R = 8, 6
        served = [[] for _ in range(S)]

        def session(sid):
            for k in range(R):
                sch.infer(sid + 1, 16 * k, 16, 1000 + sid, 2000 + sid)
                served[sid].append(k)
                time.sleep(0.002)                # the session's own work between two steps

        ts = [threading.Thread(target=session, args=(sid,)) for sid in range(S)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(timeout=30)
            assert not t.is_alive()
        sch.close()
        per_session = {}
        for call in eng.infer_calls:
            for aid, index, batch, _, _ in call:
                per_session.setdefault(aid, []).append(index)
        assert all(v == [16 * k for k in range(R)] for v in per_session.values()) and len(per_session) == S
        assert all(v == list(range(R)) for v in served)
        return eng, sch

    eng2, sch2 = run(2)
    assert eng2.peak == 2 and sch2.stats["overlapped_calls"] > 0
    assert sch2.stats["requests"] == 48 and sch2.stats["max_requests_per_call"] >= 3
    eng1, sch1 = run(1)
    assert eng1.peak == 1 and sch1.stats["overlapped_calls"] == 0 and sch1.stats["requests"] == 48


def test_lone_session_takes_the_direct_path_and_late_arrivals_still_batch(monkeypatch):
    """A lone session's call goes down in its own thread without a request object (BatchingScheduler._run_solo: the Python between two
    calls of one session is GPU idle time).  What must still hold: the in-flight accounting (a session arriving DURING a solo call
    queues and is served - behind it or overlapped -, never more than LTK_INFLIGHT calls inside the engine), errors reach the caller,
    and once multi-request batches form, the leader path with its window takes over again."""
    pytest.importorskip("torch")
    from livetalking_amd import scheduler

    class SlowEngine(FakeEngine):
        def __init__(self):
            super().__init__()
            self.inside, self.peak = 0, 0

        def wav2lip_infer(self, reqs, stream=0):
            with self.lock:
                self.inside += 1
                self.peak = max(self.peak, self.inside)
                self.infer_calls.append(list(reqs))
            if any(r[0] == 666 for r in reqs):
                with self.lock:
                    self.inside -= 1
                raise RuntimeError("bad avatar")
            time.sleep(0.05)
            with self.lock:
                self.inside -= 1

    eng = SlowEngine()
    sch = scheduler.BatchingScheduler(eng, "wav2lip")
    solo_calls = []
    orig = sch._run_solo
    monkeypatch.setattr(sch, "_run_solo", lambda args, frames: (solo_calls.append(args[0]), orig(args, frames))[1])
    for k in range(3):                                     # a lone session: every call direct, one request per call
        sch.infer(1, 16 * k, 16, 10, 20)
    assert solo_calls == [1, 1, 1] and sch.stats["calls"] == 3 and sch.stats["max_requests_per_call"] == 1
    with pytest.raises(RuntimeError, match="bad avatar"):
        sch.infer(666, 0, 16, 10, 20)
    assert sch._inflight == 0 and sch._inflight_frames == 0
    # sessions 2 and 3 arrive while session 1's solo call is inside the engine
    done = []
    t1 = threading.Thread(target=lambda: (sch.infer(1, 48, 16, 10, 20), done.append(1)))
    t1.start()
    time.sleep(0.005)
    late = [threading.Thread(target=lambda sid=sid: (sch.infer(sid, 0, 16, 10, 20), done.append(sid))) for sid in (2, 3)]
    for t in late:
        t.start()
    for t in [t1] + late:
        t.join(timeout=10)
        assert not t.is_alive()
    assert sorted(done) == [1, 2, 3] and eng.peak <= 2
    assert any(len(c) == 2 for c in eng.infer_calls)       # the two late sessions rode one call
    assert sch._inflight == 0 and sch._inflight_frames == 0
    n_solo = len(solo_calls)
    sch.infer(1, 64, 16, 10, 20)                           # a multi-request batch formed < 2 s ago: leader path (window), not solo
    assert len(solo_calls) == n_solo
    sch.close()


def test_idle_workers_exit_and_engine_close_closes_schedulers(monkeypatch):
    """A scheduler must not pin its engine for the life of the process: worker threads leave after LTK_WORKER_IDLE_S idle seconds
    (a later request starts new ones), and Engine.close() closes the schedulers stored on the engine."""
    pytest.importorskip("torch")
    from livetalking_amd import scheduler
    monkeypatch.setenv("LTK_WORKER_IDLE_S", "0.1")

    class SlowEngine(FakeEngine):
        def wav2lip_infer(self, reqs, stream=0):
            time.sleep(0.05)
            super().wav2lip_infer(reqs, stream)

    eng = SlowEngine()
    sch = scheduler.get_scheduler(eng)
    ts = [threading.Thread(target=sch.infer, args=(1, 16 * i, 16, 1, 2)) for i in range(3)]
    for t in ts:
        t.start()
        time.sleep(0.01)
    for t in ts:
        t.join(timeout=10)
    assert any(w.is_alive() for w in sch._workers)           # the queued requests were served by workers
    time.sleep(0.5)
    assert not any(w.is_alive() for w in sch._workers)       # ... which left when idle
    sch.infer(1, 64, 16, 1, 2)                               # still usable
    from livetalking_amd.engine import Engine
    closed = []
    dummy = Engine.__new__(Engine)                           # Engine.close() without a GPU: only the scheduler part runs
    dummy._closed, dummy._egress_open, dummy._h = False, set(), None
    dummy._lib = types.SimpleNamespace(ltk_engine_destroy=lambda h: closed.append(h))
    s2 = scheduler.get_scheduler(dummy) if hasattr(dummy, "wav2lip_infer") else None
    assert s2 is not None and "_ltk_schedulers" in dummy.__dict__
    dummy.close()
    assert "_ltk_schedulers" not in dummy.__dict__ and closed == [None]


def test_scheduler_soak_random_arrivals(monkeypatch):
    """Randomised soak of the scheduler (the render loop's inference threads block in infer(): a lost wake-up would hang a session for
    good): 1..33 sessions with random think times, idle gaps long enough for the workers to exit and come back, 2 % poisoned requests,
    against an engine that serialises device work like the real one.  Every request returns (or raises its own error), never more
    calls are inside the engine than LTK_INFLIGHT allows, nothing hangs."""
    pytest.importorskip("torch")
    import random
    from livetalking_amd import scheduler
    monkeypatch.setenv("LTK_WORKER_IDLE_S", "0.05")

    class Eng(FakeEngine):
        def __init__(self):
            super().__init__()
            self.gpu = threading.Lock()
            self.inside = self.peak = 0

        def wav2lip_infer(self, reqs, stream=0):
            with self.lock:
                self.inside += 1
                self.peak = max(self.peak, self.inside)
            try:
                if any(r[0] == 99 for r in reqs):
                    raise RuntimeError("bad avatar")
                with self.gpu:
                    time.sleep(2e-5 * sum(r[2] for r in reqs))
            finally:
                with self.lock:
                    self.inside -= 1

    for seed, (S, inflight) in enumerate(((1, 2), (2, 2), (5, 2), (16, 2), (16, 1), (33, 2))):
        monkeypatch.setenv("LTK_INFLIGHT", str(inflight))
        eng, steps = Eng(), 20
        sch = scheduler.BatchingScheduler(eng)
        done, errs = [0] * S, []

        def sess(i):
            rnd = random.Random(seed * 100 + i)
            for k in range(steps):
                try:
                    sch.infer(99 if rnd.random() < 0.02 else 1, k * 16, 16, 1, 2)
                except RuntimeError as ex:
                    if "bad avatar" not in str(ex):
                        errs.append(str(ex))
                done[i] += 1
                if rnd.random() < 0.3:
                    time.sleep(rnd.random() * 2e-3)
                if rnd.random() < 0.02:
                    time.sleep(0.08)                     # long enough for idle workers to leave

        ts = [threading.Thread(target=sess, args=(i,), daemon=True) for i in range(S)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(timeout=60)
            assert not t.is_alive(), f"session thread hung (S={S}, inflight={inflight}): pending {len(sch._pending)}, in flight {sch._inflight}"
        sch.close()
        assert not errs and all(d == steps for d in done) and eng.peak <= inflight, (errs, done, eng.peak)


def test_poisoned_request_fails_alone_and_close_unblocks():
    """One bad request inside a 3-request batch raises only in its own caller (the reference's inference thread has no
    try/except, base_avatar.py:366: an error delivered to every co-batched session would kill all of their threads); close()
    never leaves a queued caller blocked and the scheduler keeps working afterwards."""
    pytest.importorskip("torch")
    from livetalking_amd import scheduler

    class SlowEngine(FakeEngine):
        def wav2lip_infer(self, reqs, stream=0):
            if any(r[0] == 99 for r in reqs):
                raise RuntimeError("unknown avatar 99")
            time.sleep(0.1)
            super().wav2lip_infer(reqs, stream)

    eng = SlowEngine()
    sch = scheduler.BatchingScheduler(eng, "wav2lip")
    results = {}

    def call(tag, aid, index):
        try:
            sch.infer(aid, index, 16, 1000 + index, 2000 + index)
            results[tag] = "ok"
        except RuntimeError as ex:
            results[tag] = str(ex)

    first = threading.Thread(target=call, args=("first", 1, 0))
    first.start()
    time.sleep(0.03)                                     # in flight: the next three queue up and go down as one call
    late = [threading.Thread(target=call, args=(tag, aid, idx)) for tag, aid, idx in (("a", 1, 16), ("bad", 99, 32), ("c", 1, 48))]
    for t in late:
        t.start()
    for t in [first] + late:
        t.join(timeout=10)
        assert not t.is_alive()
    assert results == {"first": "ok", "a": "ok", "bad": "unknown avatar 99", "c": "ok"}, results
    served = sorted(r[1] for c in eng.infer_calls for r in c)
    assert served == [0, 16, 48], eng.infer_calls          # the good requests of the poisoned batch were re-issued and served
    assert scheduler.get_scheduler(eng) is scheduler.get_scheduler(eng)      # held on the engine object, not in a global table
    assert not hasattr(scheduler, "_SCHEDULERS")
    sch.close()
    sch.infer(1, 64, 16, 1, 2)                           # usable after close()
    sch.close()


def test_least_loaded_placement():
    from livetalking_amd.sharding import LeastLoaded
    p = LeastLoaded(2, capacity_per_gpu=2)
    assert [p.place(s) for s in "abcd"] == [0, 1, 0, 1]
    with pytest.raises(RuntimeError):
        p.place("e")
    p.release("a")
    assert p.place("e") == 0 and p.place("b") == 1


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="needs the upstream checkout (build container only)")
def test_plugin_module_contract_against_reference_app():
    """The names app.py pulls from the plugin module exist with the reference's
    signatures (app.py:134-151) and LipReal registers under ("avatar","wav2lip")."""
    import inspect
    import livetalking_amd.avatars.wav2lip_avatar as mine
    src = open(os.path.join(REFERENCE, "app.py")).read()
    for fn in ("load_model", "load_avatar", "warm_up"):
        assert f"avatar_mod.{fn}" in src or fn in src
        assert callable(getattr(mine, fn))
    assert list(inspect.signature(mine.warm_up).parameters)[:3] == ["batch_size", "model", "modelres"]
    assert list(inspect.signature(mine.load_model).parameters)[0] == "path"
    assert list(inspect.signature(mine.LipReal.__init__).parameters)[1:4] == ["opt", "model", "avatar"]
    assert list(inspect.signature(mine.LipReal.inference_batch).parameters)[1:] == ["index", "audiofeat_batch"]
    assert list(inspect.signature(mine.LipReal.paste_back_frame).parameters)[1:] == ["pred_frame", "idx"]


def test_host_e4m3_conversion_matches_torch():
    """The weight packer's fp32 -> OCP e4m3fn conversion (csrc/conv_mfma.hip f32_to_e4m3) against torch.float8_e4m3fn:
    round to nearest even, subnormals, saturation at +-448."""
    import ctypes as C

    import numpy as np
    import torch
    from livetalking_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(0)
    vals = np.concatenate([rng.standard_normal(20000).astype(np.float32) * s for s in (1e-3, 0.02, 1, 30, 300)] +
                          [np.array([0, 448, -448, 500, -1e9, 1e-9, 2 ** -10, 2 ** -9, 1.5 * 2 ** -9, 2.5 * 2 ** -9, 0.0155, 0.015625,
                                     464, 447.9, 240, 232, 0.0146484375], np.float32)])
    out = np.empty(vals.size, np.uint8)
    assert lib.ltk_f32_to_e4m3(vals.ctypes.data, vals.size, out.ctypes.data) == 0
    ref = torch.from_numpy(vals).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    diff = np.nonzero(out != ref)[0]
    diff = [i for i in diff if (out[i] & 0x7F) or (ref[i] & 0x7F)]          # +0 vs -0 is not a difference
    assert not diff, [(float(vals[i]), int(out[i]), int(ref[i])) for i in diff[:8]]


def test_abi_argument_validation_without_gpu():
    """Every entry point rejects NULL / out-of-range arguments with LTK_E_INVALID and a message, before touching HIP;
    creating an engine on a box without a GPU fails with an error code instead of crashing (the product path fails loudly)."""
    import ctypes as C

    import torch
    from livetalking_amd import _lib
    lib = _lib.load()
    E_INVALID = -1
    null = C.c_void_p(None)
    assert lib.ltk_paste_back(null, 0, 0, null, null, 0, null) == E_INVALID and b"bad arguments" in lib.ltk_last_error()
    assert lib.ltk_paste_blend(null, 0, 0, null, null, 0, null) == E_INVALID
    assert lib.ltk_wav2lip_infer(null, None, 0, null) < 0
    assert lib.ltk_musetalk_infer(null, None, 0, null) < 0
    assert lib.ltk_mel_step(null, null, 0, null, 0, null, null) < 0
    assert lib.ltk_whisper_step(null, null, 0, 0, 0, 0, 0, null, null) < 0
    assert lib.ltk_egress_open(null, 4, 4, C.byref(C.c_void_p())) == E_INVALID
    assert lib.ltk_egress_frame(null, null, None, null, null) == E_INVALID
    assert lib.ltk_musetalk_set_fp8(null, 1, 0.0) == E_INVALID
    assert lib.ltk_f32_to_e4m3(null, 4, null) == E_INVALID
    assert lib.ltk_conv2d_fp8(null, null, 1, 1, 1, 32, null, 32, null, null, 8.0, null, 0, null, 0, None) == E_INVALID
    if not torch.cuda.is_available():
        h = C.c_void_p()
        rc = lib.ltk_engine_create(0, C.byref(h))
        assert rc < 0 and not h.value and lib.ltk_last_error()
        from livetalking_amd.engine import Engine
        with pytest.raises(Exception):
            Engine(0)


def test_tile_table_entries_name_existing_layers_and_legal_tiles():
    """csrc/engine.hip kTileTable is keyed by layer-name strings and was tuned on single boxes: every entry must name a layer
    of the network description, a frame-count bucket and a tile / split conv3 has an instantiation for (device-free check
    inside the library, include/ltk.h ltk_debug_tile_table_check); the table itself is read from the source here so that the
    test also notices an entry the C side does not parse as expected."""
    import ctypes as C
    import re
    from livetalking_amd import _lib
    lib = _lib.load()
    buf = C.create_string_buffer(4096)
    bad = lib.ltk_debug_tile_table_check(buf, len(buf))
    assert bad == 0, buf.value.decode()
    src = open(os.path.join(ROOT, "livetalking_amd", "csrc", "engine.hip")).read()
    table = src[src.index("const TileEntry kTileTable[] = {"):]
    table = table[:table.index("};")]
    entries = re.findall(r'\{"([a-z_.0-9]+)",\s*(\d+),\s*(\d+),\s*(\d+),\s*(\d+)\}', table)
    assert len(entries) >= 10
    layers = set(re.findall(r'\{"((?:audio_encoder|face_encoder_blocks|face_decoder_blocks|output_block)[.0-9]+)",\s*(?:false|true)', src))
    assert len(layers) == 54
    for name, bucket, pxw, nbt, ks in entries:
        assert name in layers, name
        assert int(bucket) in range(5) and int(pxw) in (0, 1, 2, 4) and int(nbt) in (0, 1, 2) and 0 <= int(ks) <= 32


def test_profile_kernel_filters_agree_on_the_recorded_kernel_names():
    """bench.py's live PMC traffic sum and scripts/make_profile_summary.py's per-kernel tables select "the layer kernels of a Wav2Lip
    pass" by name; un-templated kernels are listed under their mangled names (round 6: convs2d_kernel's rows were dropped by a
    startswith("conv") test).  Both predicates over every kernel name of the committed traces: equal, every layer kernel in, helpers out."""
    import csv
    import importlib.util
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("make_profile_summary", os.path.join(root, "scripts", "make_profile_summary.py"))
    mps = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mps)
    names = set()
    for f in ("r06_w2l_kernel_stats.csv", "r06_w2l256_kernel_stats.csv"):
        with open(os.path.join(root, "profiles", f)) as fh:
            rows = [r for r in csv.reader(l for l in fh if not l.startswith("#"))]
        names |= {r[0] for r in rows[1:]}
    assert len(names) > 15
    for n in names:
        assert bench.is_layer_kernel(n) == mps.is_layer_kernel(n), n
    layer = {n for n in names if bench.is_layer_kernel(n)}
    for must in ("conv7_kernel", "conv3_kernel", "conv3_head_kernel", "convs2d_kernel", "audio0_kernel", "audio3_kernel", "rowconv_kernel", "rowgemm_kernel"):
        assert any(must in n for n in layer), must
    for n in names - layer:                  # what stays out: torch's fills, the runtime's blits, table uploads, the test-hook head
        assert "ltk" not in n or any(h in n for h in ("upload", "pack", "head_kernel", "paste", "mel")), n


def test_kernel_resources_script_follows_the_makefile():
    """scripts/kernel_resources.py / isa_spill_report.py recompile the sources device-only: their flags must be the Makefile's
    (a per-file flag such as -amdgpu-mfma-vgpr-form changes the register allocation they report)."""
    import importlib.util
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(root, "scripts", "kernel_resources.py"))
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    with open(os.path.join(root, "livetalking_amd", "csrc", "Makefile")) as f:
        mk = f.read()
    base = re.search(r"^CXXFLAGS\s*=\s*(.+)$", mk, re.M).group(1).replace("$(ARCH)", re.search(r"^ARCH\s*\?=\s*(\S+)", mk, re.M).group(1)).split()
    assert kr.BASE == base
    extra = {}
    for objs, flags in re.findall(r"^([\w. ]+\.o):\s*CXXFLAGS\s*\+=\s*(.+)$", mk, re.M):
        for o in objs.split():
            extra.setdefault(o.replace(".o", ".hip"), []).extend(flags.split())
    assert kr.EXTRA == extra
    assert set(kr.makefile_sources()) >= set(extra) and len(kr.makefile_sources()) == 10
    # the name shortener on the two kinds of names the compiler's remarks carry
    assert kr.pretty("_ZN3ltk14convs2d_kernelILi2EEEvPKDF16_ii", "_ZN3ltk14convs2d_kernelILi2EEEvPKDF16_ii") == "ltk::convs2d_kernel<2>"
    assert kr.pretty("x", "void ltk::conv7_kernel<true>(ltk::C7Args, ltk::FacePtrs const*)") == "ltk::conv7_kernel<true>"
