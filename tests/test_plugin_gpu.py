"""End-to-end through the plugin surface (module functions + LipReal) on the GPU:
PCM chunks in -> composited uint8 frames out, compared with the oracle chain
(mel_oracle -> plugin_oracle -> paste_oracle), following the call order of the
reference's render/inference/process_frames loops (avatars/base_avatar.py:337-376,
:433, :487-494)."""
import argparse
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import mel_oracle, paste_oracle, plugin_oracle  # noqa: E402
import synth_inputs as synth


@pytest.mark.gpu
def test_lipreal_headless_render_loop():
    import livetalking_amd.avatars.wav2lip_avatar as plugin
    from livetalking_amd.hostshim import mirror_index

    sd_np = synth.wav2lip_state_dict(1234)
    model = plugin.load_model(None, state_dict=sd_np, max_frames=8)
    B = 4
    plugin.warm_up(B, model, 256)
    avatar = synth.wav2lip_avatar(n_frames=5, full_hw=(360, 640), box=160, seed=0)
    frames, faces, coords = avatar
    opt = argparse.Namespace(fps=25, batch_size=B, l=10, r=10, sessionid=0)
    sess = plugin.LipReal(opt, model, avatar)
    assert sess.get_avatar_length() == 5 and hasattr(sess, "asr")

    audio = synth.synthetic_audio(2.0)
    n_steps = 3
    # warm_up consumed l+r chunks of silence (queue was empty); now feed speech
    for c in range(n_steps * 2 * B):
        sess.put_audio_frame(audio[c * 320:(c + 1) * 320], {})
    sd = {k: torch.from_numpy(v) for k, v in sd_np.items()}
    pcm_hist = [np.zeros(320, np.float32)] * 20
    index = 0
    for step in range(n_steps):
        sess.asr.run_step()                                   # render thread
        feat = sess.asr.feat_queue.get(timeout=1)             # inference thread
        audio_frames = [sess.asr.output_queue.get() for _ in range(2 * B)]
        assert len(audio_frames) == 2 * B      # audio out lags features by the r look-ahead chunks
        pred = sess.inference_batch(index, feat)
        assert len(pred) == B
        # oracle for the same step
        pcm_hist = pcm_hist + [audio[c * 320:(c + 1) * 320] for c in range(step * 2 * B, (step + 1) * 2 * B)]
        wav = np.concatenate(pcm_hist)
        ref_feats = mel_oracle.mel_chunks(wav, len(pcm_hist))
        assert float(np.abs(feat.cpu().numpy() - np.stack(ref_feats)).max()) <= 1e-3
        ref_pred = plugin_oracle.inference_batch(sd, faces, index, B, ref_feats)
        for i, res_frame in enumerate(pred):                  # process thread
            idx = mirror_index(len(frames), index + i)
            out = sess.paste_back_frame(res_frame, idx)
            assert out.dtype == np.uint8 and out.shape == (360, 640, 3) and out.flags["C_CONTIGUOUS"] and out.flags["WRITEABLE"]
            ref = paste_oracle.paste_back_frame(ref_pred[i], frames[idx], coords[idx])
            d = np.abs(out.astype(np.int32) - ref.astype(np.int32))
            assert d.max() <= 6 and (d <= 2).mean() >= 0.995, (step, i, int(d.max()))
            # bit-exact composite given the engine's own uint8 crop
            own = paste_oracle.paste_back_frame(res_frame.cpu().numpy().astype(np.float32), frames[idx], coords[idx])
            assert np.array_equal(out, own)
        index += B
        pcm_hist = pcm_hist[-20:]
    model.engine.close()


def _psnr(a, b):
    d = a.astype(np.float64) - b.astype(np.float64)
    m = float((d * d).mean())
    return 99.0 if m == 0 else 10 * np.log10(255.0 ** 2 / m)


@pytest.mark.gpu
def test_file_format_legs_checkpoint_and_avatar_dir(tmp_path, monkeypatch):
    """W9 / 8(f)-1 end to end on the GPU path: load_model(path) on a checkpoint FILE in the reference's format
    ({"state_dict": {"module.<name>": tensor}}, wav2lip_avatar.py:51-70) and load_avatar(id) on a directory in genavatar's
    layout (full_imgs/%08d.png, face_imgs/%08d.png, coords.pkl; wav2lip/genavatar.py:124-138) packed to bank.ltkbank,
    then a render step; frames must equal the ones from the in-memory state dict / in-memory bank bit for bit."""
    import pickle
    from PIL import Image
    import livetalking_amd.avatars.wav2lip_avatar as plugin
    from livetalking_amd import bank

    sd_np = synth.wav2lip_state_dict(1234)
    ckpt = {"state_dict": {"module." + k: torch.from_numpy(v) for k, v in sd_np.items()}}
    # BatchNorm's num_batches_tracked entries exist in a real checkpoint (380 tensors): they must be ignored
    for k in list(ckpt["state_dict"]):
        if k.endswith("running_var"):
            ckpt["state_dict"][k.replace("running_var", "num_batches_tracked")] = torch.tensor(1000)
    os.makedirs(tmp_path / "models")
    path = str(tmp_path / "models" / "wav2lip.pth")
    torch.save(ckpt, path)

    frames, faces, coords = synth.wav2lip_avatar(n_frames=4, full_hw=(180, 320), box=96, seed=5)
    adir = tmp_path / "data" / "avatars" / "avt1"
    os.makedirs(adir / "full_imgs"); os.makedirs(adir / "face_imgs")
    for i in range(4):
        Image.fromarray(np.ascontiguousarray(frames[i][..., ::-1])).save(adir / "full_imgs" / f"{i:08d}.png")     # cv2.imwrite stores BGR arrays as RGB files
        Image.fromarray(np.ascontiguousarray(faces[i][..., ::-1])).save(adir / "face_imgs" / f"{i:08d}.png")
    with open(adir / "coords.pkl", "wb") as f:
        pickle.dump(coords, f)
    bank.pack_avatar_dir(str(adir))
    monkeypatch.chdir(tmp_path)                                     # load_avatar reads ./data/avatars/<id>
    monkeypatch.setenv("LTK_DEVICES", "0")

    model = plugin.load_model(path, max_frames=8)                   # the torch.load(path)["state_dict"] leg
    avatar = plugin.load_avatar("avt1")                             # the .ltkbank leg
    assert len(avatar[0]) == 4 and all(np.array_equal(a, b) for a, b in zip(avatar[1], faces))
    B = 4
    opt = argparse.Namespace(fps=25, batch_size=B, l=10, r=10, sessionid=0)
    sess = plugin.LipReal(opt, model, avatar)
    feat = torch.from_numpy(np.random.default_rng(0).standard_normal((B, 80, 16)).astype(np.float32)).cuda()
    got = torch.stack(sess.inference_batch(2, feat)).cpu().numpy()
    outs = [sess.paste_back_frame(sess.inference_batch(2, feat)[i], plugin.mirror_index(4, 2 + i)) for i in range(B)]

    model2 = plugin.load_model(None, state_dict=sd_np, max_frames=8)
    sess2 = plugin.LipReal(opt, model2, (frames, faces, coords))
    ref = torch.stack(sess2.inference_batch(2, feat)).cpu().numpy()
    assert np.array_equal(got, ref)
    for i in range(B):
        idx = plugin.mirror_index(4, 2 + i)
        assert np.array_equal(outs[i], sess2.paste_back_frame(sess2.inference_batch(2, feat)[i], idx))
    for m in (model, model2):
        for e in m.engines:
            e.close()


@pytest.mark.gpu
def test_two_engines_two_sessions_render_concurrently(monkeypatch):
    """8(e) inside one process: LTK_DEVICES=0,0 builds two engines (here on the same GPU); two sessions land on different
    engines and render the same request concurrently from two threads: identical frames, each engine touched once."""
    import threading
    import livetalking_amd.avatars.wav2lip_avatar as plugin
    monkeypatch.setenv("LTK_DEVICES", "0,0")
    sd_np = synth.wav2lip_state_dict(1234)
    model = plugin.load_model(None, state_dict=sd_np, max_frames=8)
    assert len(model.engines) == 2 and model.engines[0] is not model.engines[1]
    plugin.warm_up(4, model, 256)
    avatar = synth.wav2lip_avatar(n_frames=5, full_hw=(360, 640), box=160, seed=0)
    B = 4
    opt = argparse.Namespace(fps=25, batch_size=B, l=10, r=10, sessionid=0)
    s0, s1 = plugin.LipReal(opt, model, avatar), plugin.LipReal(opt, model, avatar)
    assert (s0._slot, s1._slot) == (0, 1) and s0.engine is not s1.engine
    feat = torch.from_numpy(np.random.default_rng(1).standard_normal((B, 80, 16)).astype(np.float32)).cuda()
    res = {}

    def run(name, s):
        frames = []
        for step in range(3):
            pred = s.inference_batch(step * B, feat)
            frames += [s.paste_back_frame(pred[i], plugin.mirror_index(5, step * B + i)) for i in range(B)]
        res[name] = np.stack(frames)

    ts = [threading.Thread(target=run, args=("a", s0)), threading.Thread(target=run, args=("b", s1))]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=120)
    assert "a" in res and "b" in res and np.array_equal(res["a"], res["b"])
    assert s0._sched.stats["calls"] == 3 and s1._sched.stats["calls"] == 3 and s0._sched is not s1._sched
    for e in model.engines:
        e.close()


@pytest.mark.gpu
def test_pool_of_four_engines_32_sessions_equals_single_engine(monkeypatch):
    """The reference deploys ONE process (app.py:62-63,99; server/session_manager.py:56-94): on an N-GPU node that process holds an
    EnginePool of N engines.  Here N = 4 (all on GPU 0: the box has one), 8 sessions per engine, 32 session threads rendering
    three steps each through inference_batch + paste_back_frame at once: the placement is 8 / 8 / 8 / 8, every engine's scheduler
    coalesces only its own sessions, and every session's composited frames are byte-identical to the same session rendered alone
    on a single engine.  (LTK_SPLITK=0: one summation order per output element whatever a call was coalesced with - without it
    frames differ by <= 1 LSB with the coalescing pattern, which is timing dependent.)"""
    import threading
    import livetalking_amd.avatars.wav2lip_avatar as plugin
    from livetalking_amd.engine import Engine
    E, per, B, steps = 4, 8, 4, 3
    sd_np = synth.wav2lip_state_dict(1234)
    avatars = [synth.wav2lip_avatar(n_frames=5, full_hw=(180, 320), box=96, seed=20 + k) for k in range(3)]
    rng = np.random.default_rng(7)
    feats = [torch.from_numpy(rng.standard_normal((B, 80, 16)).astype(np.float32)).cuda() for _ in range(E * per)]

    def render(sess, i, out):
        frames = []
        for step in range(steps):
            index = step * B + i
            pred = sess.inference_batch(index, feats[i])
            frames += [sess.paste_back_frame(pred[k], plugin.mirror_index(5, index + k)).copy() for k in range(B)]
        out[i] = np.stack(frames)

    Engine.set_knob("SPLITK", 0)
    try:
        # reference run: one engine, the sessions one after the other
        monkeypatch.setenv("LTK_DEVICES", "0")
        model1 = plugin.load_model(None, state_dict=sd_np, max_frames=64)
        ref = {}
        for i in range(E * per):
            opt = argparse.Namespace(fps=25, batch_size=B, l=10, r=10, sessionid=i)
            render(plugin.LipReal(opt, model1, avatars[i % 3]), i, ref)
        for e in model1.engines:
            e.close()
        # the pool: 4 engines, 32 sessions at once
        monkeypatch.setenv("LTK_DEVICES", "0,0,0,0")
        model = plugin.load_model(None, state_dict=sd_np, max_frames=64)
        assert len(model.engines) == E and len({id(e) for e in model.engines}) == E
        sessions = [plugin.LipReal(argparse.Namespace(fps=25, batch_size=B, l=10, r=10, sessionid=i), model, avatars[i % 3]) for i in range(E * per)]
        assert sorted(s._slot for s in sessions) == sorted(list(range(E)) * per), "least-loaded placement: 8 sessions per engine"
        got = {}
        ts = [threading.Thread(target=render, args=(sessions[i], i, got)) for i in range(E * per)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(timeout=180)
        assert len(got) == E * per
        for i in range(E * per):
            assert np.array_equal(got[i], ref[i]), f"session {i} (engine {sessions[i]._slot})"
        scheds = {id(s._sched): s._sched for s in sessions}
        assert len(scheds) == E
        tot = sum(sc.stats["requests"] for sc in scheds.values())
        print("[pool] per-engine scheduler stats: " + "; ".join(f"calls {sc.stats['calls']} requests {sc.stats['requests']} max/call {sc.stats['max_requests_per_call']}" for sc in scheds.values()))
        assert tot == E * per * steps and all(sc.stats["requests"] == per * steps for sc in scheds.values())
        for e in model.engines:
            e.close()
    finally:
        Engine.set_knob("SPLITK", 1)


@pytest.mark.gpu
def test_paste_back_batch_equals_per_frame_and_frames_stay_valid():
    """LipReal.paste_back_frame through the batch path (B composites on the device + ONE pinned device-to-host copy on the first
    request of a batch: ltk_paste_back_batch) returns, frame by frame, exactly what the per-frame entry point returns, the
    arrays are writable C-contiguous (H,W,3) uint8 as base_avatar.py:449-452 needs, and a frame the caller keeps (or draws on)
    is not touched by later batches."""
    import argparse
    import livetalking_amd.avatars.wav2lip_avatar as plugin
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    B = 6
    model = plugin.load_model(None, state_dict=synth.wav2lip_state_dict(1234), max_frames=B, device=0)
    try:
        avatar = synth.wav2lip_avatar(n_frames=4, full_hw=(360, 640), box=160, seed=0)
        opt = argparse.Namespace(fps=25, batch_size=B, l=10, r=10, sessionid=0)
        sess = plugin.LipReal(opt, model, avatar)
        feat = torch.from_numpy(np.random.default_rng(3).standard_normal((B, 80, 16)).astype(np.float32)).cuda()
        index = 2                                       # bank frames 2,3,3,2,1,0: across the ping-pong turn
        items = sess.inference_batch(index, feat)
        got = [sess.paste_back_frame(items[i], plugin.mirror_index(4, index + i)) for i in range(B)]
        for i in range(B):
            ref = np.empty((360, 640, 3), np.uint8)
            sess.engine.paste_back(sess._aid, plugin.mirror_index(4, index + i), items[i].data_ptr(), ref)
            assert got[i].dtype == np.uint8 and got[i].shape == (360, 640, 3) and got[i].flags["C_CONTIGUOUS"] and got[i].flags["WRITEABLE"]
            assert np.array_equal(got[i], ref), i
        keep = got[1].copy()
        got[2][:40, :200] = 7                           # the caller draws on ITS frame (cv2.putText in the reference)
        for step in range(3):                           # later batches allocate / reuse pinned blocks
            more = sess.inference_batch(index + (step + 1) * B, feat)
            [sess.paste_back_frame(more[i], plugin.mirror_index(4, index + (step + 1) * B + i)) for i in range(B)]
        assert np.array_equal(got[1], keep) and (got[2][:40, :200] == 7).all()
        # an index that is not the batch's own falls back to the per-frame path and still composites the asked frame
        other = sess.paste_back_frame(items[0], 0)
        ref0 = np.empty((360, 640, 3), np.uint8)
        sess.engine.paste_back(sess._aid, 0, items[0].data_ptr(), ref0)
        assert np.array_equal(other, ref0)
    finally:
        for e in model.engines:
            e.close()


@pytest.mark.gpu
def test_concurrent_sessions_two_calls_in_flight_bit_identical():
    """The defaults of round 4 together on ONE engine: continuous batching with up to two engine calls in flight (scheduler.py) over
    passes replayed from captured hipGraphs (knob GRAPH).  Six session threads run five steps each, free-running, each at its own
    bank position with its own mel windows; whatever they were batched with and whether their call was issued while another was
    still running, every session gets byte for byte the frames it gets alone (LTK_SPLITK=0: one summation order per output element
    whatever the launch's frame count).  The threaded phase is repeated (up to 4 rounds) until the scheduler's own counter says
    that at least one call was issued while another was in flight (`overlapped_calls`), so the two-calls-in-flight path is
    known to be exercised."""
    import argparse
    import threading
    import livetalking_amd.avatars.wav2lip_avatar as plugin
    from livetalking_amd.engine import Engine
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    S, B, STEPS = 6, 4, 5
    Engine.set_knob("SPLITK", 0)
    model = plugin.load_model(None, state_dict=synth.wav2lip_state_dict(1234), max_frames=S * B, device=0)
    try:
        avatar = synth.wav2lip_avatar(n_frames=5, full_hw=(360, 640), box=160, seed=0)
        sessions = [plugin.LipReal(argparse.Namespace(fps=25, batch_size=B, l=10, r=10, sessionid=s), model, avatar) for s in range(S)]
        rng = np.random.default_rng(11)
        feats = [torch.from_numpy(rng.standard_normal((B, 80, 16)).astype(np.float32)).cuda() for _ in range(S)]
        alone = []
        for s in range(S):                                   # reference: every session on its own, one call at a time
            alone.append([torch.stack(sessions[s].inference_batch(3 * s + step * B, feats[s])).cpu() for step in range(STEPS)])
        rounds = 0
        st = sessions[0]._sched.stats
        while rounds < 4 and (rounds == 0 or st["overlapped_calls"] == 0):
            rounds += 1
            together = [[None] * STEPS for _ in range(S)]
            go = threading.Barrier(S)

            def run(s):
                go.wait()
                for step in range(STEPS):
                    together[s][step] = torch.stack(sessions[s].inference_batch(3 * s + step * B, feats[s])).cpu()

            ts = [threading.Thread(target=run, args=(s,)) for s in range(S)]
            for t in ts:
                t.start()
            for t in ts:
                t.join(timeout=120)
                assert not t.is_alive()
            for s in range(S):
                for step in range(STEPS):
                    assert torch.equal(together[s][step], alone[s][step]), (rounds, s, step)
        print(f"[in flight] {st}; rounds {rounds}; graphs captured: {model.engine.graph_count()}")
        assert st["requests"] == S * STEPS * (1 + rounds) and st["max_requests_per_call"] >= 2
        assert st["overlapped_calls"] > 0, "no call was ever issued while another was in flight"
        assert model.engine.graph_count() >= 1
    finally:
        Engine.set_knob("SPLITK", 1)
        for e in model.engines:
            e.close()
