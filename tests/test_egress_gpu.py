"""HIP frame egress (ltk_egress_*) vs oracle/egress_oracle.py, bit-exact: composite -> transition blend -> watermark ->
BGR24 / I420, for the Wav2Lip and MuseTalk composites, the silent path, custom host frames, vector and byte kernels."""
import argparse
import threading
import time

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import egress_oracle as eo  # noqa: E402
from oracle import paste_oracle  # noqa: E402
import synth_inputs as synth


def _wm(seed=0):
    rng = np.random.default_rng(seed)
    return (rng.integers(0, 2, (9, 61), dtype=np.uint8) * 255).astype(np.uint8), 10, 12, (128, 128, 128)


@pytest.mark.gpu
@pytest.mark.parametrize("hw", [(360, 640), (122, 202)])        # W % 4 == 0: dword kernel; 202: byte kernel
def test_egress_wav2lip_sources_formats_transition(engine, hw):
    from livetalking_amd import egress
    frames, faces, coords = synth.wav2lip_avatar(n_frames=4, full_hw=hw, box=96 if hw[0] < 200 else 160, seed=3)
    aid = engine.register_avatar(faces, frames, coords)
    rng = np.random.default_rng(1)
    preds = rng.integers(0, 256, (4, 256, 256, 3), dtype=np.uint8)
    d_preds = torch.from_numpy(preds).cuda()
    H, W = hw
    mask, wx, wy, col = _wm()
    for fmt in ("bgr24", "i420"):
        for chroma in (0, 1):
            h = engine.egress_open(H, W)
            engine.egress_watermark(h, mask, wx, wy, col)
            fcode = egress.FMT_I420 if fmt == "i420" else egress.FMT_BGR24
            cache = {False: None, True: None}
            # (speaking, idx, alpha): silent, switch to speaking with blends, back to silent, a custom frame
            seq = [(False, 0, -1.0), (True, 1, 0.0), (True, 2, 0.37), (True, 3, 0.999), (True, 0, -1.0), (False, 1, 0.25),
                   (False, 2, 0.5), ("custom", 0, 0.75)]
            for speaking, idx, alpha in seq:
                out = np.empty((H * 3 // 2, W) if fmt == "i420" else (H, W, 3), dtype=np.uint8)
                if speaking == "custom":
                    custom = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
                    engine.egress_frame(h, out, egress.SRC_HOST, 0, 0, 0, custom, False, alpha, True, fcode, chroma)
                    frame, spk = custom, False
                elif speaking:
                    engine.egress_frame(h, out, egress.SRC_WAV2LIP, aid, idx, d_preds[idx].data_ptr(), None, True, alpha, True, fcode, chroma)
                    frame, spk = paste_oracle.paste_back_frame(preds[idx].astype(np.float32), frames[idx], coords[idx]), True
                else:
                    engine.egress_frame(h, out, egress.SRC_WAV2LIP, aid, idx, 0, None, False, alpha, True, fcode, chroma)
                    frame, spk = frames[idx], False
                other = cache[not spk]
                if 0 <= alpha < 1 and other is not None:
                    frame = eo.add_weighted_u8(other, 1 - alpha, frame, alpha)
                cache[spk] = frame.copy()
                ref = eo.egress_frame(frame, (mask, wx, wy, col), fmt, chroma)
                assert np.array_equal(out, ref), (fmt, chroma, speaking, idx, alpha, int((out != ref).sum()))
            engine.egress_close(h)
    # keep=False never fills the caches: a later alpha has nothing to blend with
    h = engine.egress_open(H, W)
    out = np.empty((H, W, 3), dtype=np.uint8)
    engine.egress_frame(h, out, egress.SRC_WAV2LIP, aid, 0, 0, None, False, -1.0, False, egress.FMT_BGR24, 1)
    engine.egress_frame(h, out, egress.SRC_WAV2LIP, aid, 1, d_preds[1].data_ptr(), None, True, 0.3, False, egress.FMT_BGR24, 1)
    assert np.array_equal(out, paste_oracle.paste_back_frame(preds[1].astype(np.float32), frames[1], coords[1]))
    # errors: odd size for I420, wrong avatar size, bad index
    with pytest.raises(Exception):
        engine.egress_frame(h, out, egress.SRC_WAV2LIP, aid, 99, 0, None, False, -1.0, False, egress.FMT_BGR24, 1)
    engine.egress_close(h)
    h2 = engine.egress_open(H + 2, W)
    with pytest.raises(Exception):
        engine.egress_frame(h2, np.empty((H + 2, W, 3), np.uint8), egress.SRC_WAV2LIP, aid, 0, 0, None, False, -1.0, False, 0, 1)
    engine.egress_close(h2)
    h3 = engine.egress_open(5, 8)
    with pytest.raises(Exception):
        engine.egress_frame(h3, np.empty((8, 8), np.uint8), egress.SRC_HOST, 0, 0, 0, np.zeros((5, 8, 3), np.uint8), False, -1.0, False, egress.FMT_I420, 1)
    engine.egress_close(h3)
    engine.release_avatar(aid)


@pytest.mark.gpu
def test_egress_musetalk_composite(engine):
    from livetalking_amd import egress
    H, W = 360, 640
    n = 3
    frames, masks, coords, crop_boxes = _mt_bank(n, H, W)
    lat = [np.zeros((1, 8, 32, 32), np.float32) for _ in range(n)]
    aid = engine.register_musetalk_avatar(lat, frames, coords, masks, crop_boxes)
    rng = np.random.default_rng(2)
    preds = rng.integers(0, 256, (n, 256, 256, 3), dtype=np.uint8)
    d_preds = torch.from_numpy(preds).cuda()
    h = engine.egress_open(H, W)
    for i in range(n):
        out = np.empty((H * 3 // 2, W), dtype=np.uint8)
        engine.egress_frame(h, out, egress.SRC_MUSETALK, aid, i, d_preds[i].data_ptr(), None, True, -1.0, False, egress.FMT_I420, 1)
        comp = paste_oracle.paste_blend_frame(preds[i], frames[i], coords[i], masks[i], crop_boxes[i])
        assert np.array_equal(out, eo.bgr_to_i420(comp, 1))
        out2 = np.empty((H, W, 3), dtype=np.uint8)
        engine.egress_frame(h, out2, egress.SRC_MUSETALK, aid, i, 0, None, False, -1.0, False, egress.FMT_BGR24, 1)
        assert np.array_equal(out2, frames[i])
    engine.egress_close(h)


def _mt_bank(n, H, W):
    rng = np.random.default_rng(7)
    frames = [rng.integers(0, 256, (H, W, 3), dtype=np.uint8) for _ in range(n)]
    coords, crops, masks = [], [], []
    for i in range(n):
        x1, y1 = 200 + 3 * i, 80 + 2 * i
        x2, y2 = x1 + 150 + i, y1 + 170 - i
        xs, ys, xe, ye = x1 - 30, y1 - 40, x2 + 35, y2 + 45
        coords.append((x1, y1, x2, y2))
        crops.append((xs, ys, xe, ye))
        g = rng.integers(0, 256, (ye - ys, xe - xs), dtype=np.uint8)
        masks.append(np.repeat(g[:, :, None], 3, axis=2))
    return frames, masks, coords, crops


class _Sink:
    def __init__(self):
        self.video, self.audio, self.started, self.stopped = [], [], False, False

    def start(self):
        self.started = True

    def push_video_frame(self, f):
        self.video.append(f)

    def push_audio_frame(self, pcm, userdata=None):
        self.audio.append(pcm)

    def stop(self):
        self.stopped = True


@pytest.mark.gpu
def test_lipreal_device_process_frames_loop():
    """opt.egress="i420": the plugin's process_frames (egress.py) consumes res_frame_queue exactly like
    base_avatar.py:384-460 and pushes I420 frames; silent batches take the bank frame, speaking ones the composite."""
    import livetalking_amd.avatars.wav2lip_avatar as plugin
    from livetalking_amd.hostshim import AudioFrameData
    sd_np = synth.wav2lip_state_dict(1234)
    model = plugin.load_model(None, state_dict=sd_np, max_frames=8)
    B = 4
    avatar = synth.wav2lip_avatar(n_frames=5, full_hw=(360, 640), box=160, seed=0)
    frames, faces, coords = avatar
    opt = argparse.Namespace(fps=25, batch_size=B, l=10, r=10, sessionid=0, egress="i420", enable_transition=False)
    sess = plugin.LipReal(opt, model, avatar)
    sess.output = _Sink()
    quit_event = threading.Event()
    th = threading.Thread(target=sess.process_frames, args=(quit_event,))
    th.start()
    rng = np.random.default_rng(4)
    preds = torch.from_numpy(rng.integers(0, 256, (B, 256, 256, 3), dtype=np.uint8)).cuda()
    pcm = np.zeros(320, np.float32)
    speak = [AudioFrameData(pcm, 0, {}), AudioFrameData(pcm, 0, {})]
    quiet = [AudioFrameData(pcm, 1, {}), AudioFrameData(pcm, 1, {})]
    sess.res_frame_queue.put((None, quiet, 2))
    for i in range(B):
        sess.res_frame_queue.put((preds[i], speak, i))
    t0 = time.time()
    while len(sess.output.video) < B + 1 and time.time() - t0 < 30:
        time.sleep(0.01)
    quit_event.set()
    th.join(timeout=10)
    assert not th.is_alive() and sess.output.started and sess.output.stopped
    assert len(sess.output.video) == B + 1 and len(sess.output.audio) == 2 * (B + 1)
    wm = None                                           # no OpenCV on the test box -> no watermark bitmap
    assert np.array_equal(np.asarray(sess.output.video[0]), eo.bgr_to_i420(frames[2], 1))
    for i in range(B):
        comp = paste_oracle.paste_back_frame(preds[i].cpu().numpy().astype(np.float32), frames[i], coords[i])
        got = sess.output.video[1 + i]
        assert got.width == 640 and got.height == 360
        assert np.array_equal(np.asarray(got), eo.egress_frame(comp, wm, "i420", 1))


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["bgr24", "i420"])
def test_egress_batch_equals_per_frame_and_oracle(engine, fmt):
    """ltk_egress_batch (the speaking frames of one inference_batch result: n composites + watermark + format conversion on the
    device, ONE device-to-host copy) is, frame by frame, what ltk_egress_frame delivers and what the oracle computes, bit for bit;
    through DeviceEgress.speaking_frame_of the frames of a batch come out of one pinned block and stay valid afterwards."""
    from livetalking_amd import egress
    H, W = 360, 640
    frames, faces, coords = synth.wav2lip_avatar(n_frames=4, full_hw=(H, W), box=160, seed=3)
    aid = engine.register_avatar(faces, frames, coords)
    rng = np.random.default_rng(4)
    n = 6
    preds = rng.integers(0, 256, (n, 256, 256, 3), dtype=np.uint8)
    d_preds = torch.from_numpy(preds).cuda()
    idx = [2, 3, 3, 2, 1, 0]
    mask, wx, wy, col = _wm(2)
    fcode = egress.FMT_I420 if fmt == "i420" else egress.FMT_BGR24
    h = engine.egress_open(H, W)
    engine.egress_watermark(h, mask, wx, wy, col)
    shape = (n, H * 3 // 2, W) if fmt == "i420" else (n, H, W, 3)
    host = torch.empty(shape, dtype=torch.uint8, pin_memory=True)
    engine.egress_batch(h, egress.SRC_WAV2LIP, aid, idx, d_preds.data_ptr(), host.data_ptr(), fcode, 1)
    got = host.numpy()
    for i in range(n):
        one = np.empty(shape[1:], dtype=np.uint8)
        engine.egress_frame(h, one, egress.SRC_WAV2LIP, aid, idx[i], d_preds[i].data_ptr(), None, True, -1.0, False, fcode, 1)
        ref = eo.egress_frame(paste_oracle.paste_back_frame(preds[i].astype(np.float32), frames[idx[i]], coords[idx[i]]), (mask, wx, wy, col), fmt, 1)
        assert np.array_equal(got[i], one) and np.array_equal(got[i], ref), (fmt, i)
    engine.egress_close(h)
    # the plugin-side path: items of a batch carrying their FrameGroup
    eg = egress.DeviceEgress(engine, H, W, egress.SRC_WAV2LIP, aid, fmt=fmt, watermark=(mask, wx, wy))
    items = [d_preds[i] for i in range(n)]
    egress.FrameGroup.attach(items, d_preds, idx)
    outs = [eg.speaking_frame_of(items[i], idx[i]) for i in range(n)]
    keep = outs[1].copy()
    more = [d_preds[i] for i in range(n)]
    egress.FrameGroup.attach(more, d_preds, idx)
    [eg.speaking_frame_of(more[i], idx[i]) for i in range(n)]
    for i in range(n):
        assert np.array_equal(np.asarray(outs[i]), got[i]), i
        if fmt == "i420":
            assert isinstance(outs[i], egress.I420Frame) and (outs[i].height, outs[i].width) == (H, W)
    assert np.array_equal(np.asarray(outs[1]), keep)
    eg.close()
    engine.release_avatar(aid)


@pytest.mark.gpu
def test_engine_closed_before_its_sessions_fails_cleanly():
    """A session's finalizer (MuseReal's batched-paste egress, a DeviceEgress held by a render thread) may run after the
    engine was closed: the late calls must come back as LtkError / no-ops, not touch the freed engine, and must not
    leave a stale HIP error behind for the next engine's first launch on this thread."""
    from livetalking_amd._lib import LtkError
    from livetalking_amd.egress import SRC_HOST, DeviceEgress
    from livetalking_amd.engine import Engine

    sd = synth.wav2lip_state_dict(5)
    e1 = Engine(0)
    e1.load_wav2lip(sd, max_frames=4)
    eg = DeviceEgress(e1, 64, 96, SRC_HOST, 0, fmt="bgr24", watermark=None)
    e1.close()
    eg.close()                      # no-op: the engine closed the session with itself
    eg.close()
    with pytest.raises(LtkError):
        e1.sync()
    with pytest.raises(LtkError):
        e1.egress_open(64, 96)

    e2 = Engine(0)                  # the next engine's first launches see a clean error state
    e2.load_wav2lip(sd, max_frames=4)
    mel = np.zeros((2, 80, 16), np.float32)
    face = np.zeros((2, 6, 256, 256), np.float32)
    out = e2.wav2lip_forward_host(mel, face)
    assert out.shape[0] == 2
    e2.close()
