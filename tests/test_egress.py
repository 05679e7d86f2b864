"""CPU: the egress oracle against the BT.601 limited-range known answers, the transition protocol of
avatars/base_avatar.py:384-447, and the host-side DeviceEgress clock logic (engine faked)."""
import numpy as np

from oracle import egress_oracle as eo


def test_i420_known_answers():
    # ITU-R BT.601 limited range (studio swing) 8-bit values of the primaries
    cases = {(255, 255, 255): (235, 128, 128), (0, 0, 0): (16, 128, 128), (0, 0, 255): (81, 90, 240),
             (0, 255, 0): (145, 54, 34), (255, 0, 0): (41, 240, 110), (128, 128, 128): (126, 128, 128)}
    for bgr, (y, u, v) in cases.items():
        f = np.empty((4, 4, 3), np.uint8)
        f[:] = bgr
        for chroma in (0, 1):
            out = eo.bgr_to_i420(f, chroma)
            assert out.shape == (6, 4)
            flat = out.reshape(-1)
            assert set(flat[:16]) == {y} and set(flat[16:20]) == {u} and set(flat[20:24]) == {v}, (bgr, flat)


def test_i420_chroma_modes_and_layout():
    rng = np.random.default_rng(0)
    f = rng.integers(0, 256, (6, 8, 3), dtype=np.uint8)
    o0, o1 = eo.bgr_to_i420(f, 0), eo.bgr_to_i420(f, 1)
    assert np.array_equal(o0[:6], o1[:6])                       # luma does not depend on the chroma mode
    # chroma=0 is the top-left pixel of each quad
    sub = f[0::2, 0::2].astype(np.int64)
    u = ((eo.RU * sub[..., 2] + eo.GU * sub[..., 1] + eo.BU * sub[..., 0] + 16384) >> 15) + 128
    assert np.array_equal(o0.reshape(-1)[48:60], u.reshape(-1).astype(np.uint8))
    # a frame constant over every 2x2 quad gives the same chroma in both modes
    g = np.repeat(np.repeat(f[:3, :4], 2, axis=0), 2, axis=1)
    assert np.array_equal(eo.bgr_to_i420(g, 0), eo.bgr_to_i420(g, 1))


def test_add_weighted_rounding():
    a = np.array([[0, 1, 2, 255, 255]], np.uint8)
    b = np.array([[1, 2, 3, 255, 0]], np.uint8)
    # 0.5/0.5: exact ties round half to even like cvRound
    assert eo.add_weighted_u8(a, 0.5, b, 0.5).tolist() == [[0, 2, 2, 255, 128]]
    assert np.array_equal(eo.add_weighted_u8(a, 0.0, b, 1.0), b)
    assert np.array_equal(eo.add_weighted_u8(a, 1.0, b, 0.0), a)


def test_watermark_clipped():
    f = np.zeros((8, 8, 3), np.uint8)
    m = np.ones((3, 3), np.uint8)
    eo.apply_watermark(f, m, 6, -1, (1, 2, 3))
    assert f[0, 6].tolist() == [1, 2, 3] and f[1, 7].tolist() == [1, 2, 3] and int((f.sum(axis=2) > 0).sum()) == 4


class _Clock:
    def __init__(self):
        self.t = 100.0

    def __call__(self):
        return self.t


def test_transition_state_follows_reference_protocol():
    clk = _Clock()
    st = eo.TransitionState(True, 0.1, clk)
    sil = np.full((2, 2, 3), 10, np.uint8)
    spk = np.full((2, 2, 3), 200, np.uint8)
    # first silent frames: no speaking frame cached yet -> target frame
    clk.t += 1.0
    assert np.array_equal(st.step(sil, False), sil)
    # silent -> speaking: the clock restarts at the state change, alpha = 0 on the first frame
    clk.t += 1.0
    assert np.array_equal(st.step(spk, True), sil)                   # 1.0*last_silent + 0.0*current
    clk.t += 0.04
    out = st.step(spk, True)
    # the cached silent frame is blended with weight 1 - 0.4
    assert np.array_equal(out, eo.add_weighted_u8(sil, 1 - 0.04 / 0.1, spk, 0.04 / 0.1))
    clk.t += 0.2
    assert np.array_equal(st.step(spk, True), spk)                   # window over
    # speaking -> silent blends from the last *combined* speaking frame
    clk.t += 0.04
    first = st.step(sil, False)
    assert np.array_equal(first, spk)                                # alpha 0 at the switch
    off = eo.TransitionState(False, 0.1, clk)
    assert off.step(spk, True) is spk


class _FakeEngine:
    def __init__(self):
        self.calls = []

    def egress_open(self, H, W):
        return 7

    def egress_close(self, h):
        self.calls.append(("close", h))

    def egress_watermark(self, h, mask, x, y, color):
        self.calls.append(("wm", mask.shape, x, y, color))

    def egress_frame(self, h, out, source, avatar_id, idx, d_pred, h_frame, speaking, alpha, keep, fmt, chroma):
        self.calls.append(("frame", source, avatar_id, idx, d_pred, h_frame is not None, speaking, alpha, keep, fmt))
        return out


def test_device_egress_clock_and_requests():
    from livetalking_amd import egress
    clk = _Clock()
    fe = _FakeEngine()
    eg = egress.DeviceEgress(fe, 4, 8, egress.SRC_WAV2LIP, 3, fmt="i420", enable_transition=True, watermark=(np.ones((2, 2), np.uint8), 1, 1),
                             clock=clk)
    assert fe.calls[0][0] == "wm"
    out = eg.silent_frame(2)
    assert isinstance(out, egress.I420Frame) and out.shape == (6, 8) and [p.shape for p in out.planes()] == [(4, 8), (2, 4), (2, 4)]
    # no state change (starts silent): the clock keeps running from construction, window long over after 1 s
    clk.t += 1.0
    eg.silent_frame(2)
    assert fe.calls[-1][7] == -1.0
    clk.t += 0.5
    eg.speaking_frame(1234, 0)                                       # state change -> alpha 0
    c = fe.calls[-1]
    assert c[1:5] == (egress.SRC_WAV2LIP, 3, 0, 1234) and c[6] is True and c[7] == 0.0 and c[8] is True and c[9] == egress.FMT_I420
    clk.t += 0.05
    eg.speaking_frame(1234, 1)
    assert abs(fe.calls[-1][7] - 0.5) < 1e-9
    clk.t += 0.06
    eg.speaking_frame(1234, 2)
    assert fe.calls[-1][7] == -1.0
    custom = np.zeros((4, 8, 3), np.uint8)
    eg.silent_frame(0, custom)
    assert fe.calls[-1][1] == egress.SRC_HOST and fe.calls[-1][5] is True and fe.calls[-1][7] == 0.0
    eg.close()
    assert fe.calls[-1] == ("close", 7)
    # transition disabled: alpha is always "no blend" and nothing is cached
    eg2 = egress.DeviceEgress(fe, 4, 8, egress.SRC_MUSETALK, 1, fmt="bgr24", enable_transition=False, watermark=None, clock=clk)
    eg2.speaking_frame(1, 0)
    assert fe.calls[-1][7] == -1.0 and fe.calls[-1][8] is False and eg2._out().shape == (4, 8, 3)


class _FakePred:
    def __init__(self, ptr):
        self._p = ptr

    def data_ptr(self):
        return self._p


class _FakeModel:
    def __init__(self, engine):
        self.engine = engine


class _Out:
    def __init__(self):
        self.video, self.audio, self.started, self.stopped = [], [], False, False

    def start(self):
        self.started = True

    def push_video_frame(self, f):
        self.video.append(f)

    def push_audio_frame(self, pcm, userdata=None):
        self.audio.append((pcm.dtype, pcm.shape, userdata))

    def stop(self):
        self.stopped = True


def test_process_frames_mixin_control_flow():
    """The opt.egress process_frames loop (livetalking_amd/egress.py) against the control flow of
    avatars/base_avatar.py:384-460: silent frames take the bank frame or the custom-action cycle (with its own mirror
    index), speaking frames go through the composite, a failing frame is logged and dropped, audio is pushed as int16."""
    import argparse
    import queue
    import threading

    from livetalking_amd import egress
    from livetalking_amd.hostshim import AudioFrameData

    fe = _FakeEngine()
    calls = fe.calls

    def egress_frame(h, out, source, avatar_id, idx, d_pred, h_frame, speaking, alpha, keep, fmt, chroma):
        if d_pred == 666:
            raise RuntimeError("boom")
        calls.append(("frame", source, idx, d_pred, None if h_frame is None else int(h_frame[0, 0, 0]), speaking))
        return out

    fe.egress_frame = egress_frame

    class Sess(egress.DeviceEgressMixin):
        _egress_source = egress.SRC_MUSETALK

        def __init__(self):
            self.opt = argparse.Namespace(egress="bgr24", enable_transition=False)
            self.model = _FakeModel(fe)
            self._aid = 5
            self.frame_list_cycle = [np.zeros((4, 8, 3), np.uint8)] * 3
            self.res_frame_queue = queue.Queue()
            self.output = _Out()
            self.custom_index = {2: 0}
            self.custom_img_cycle = {2: [np.full((4, 8, 3), 10 + i, np.uint8) for i in range(2)]}
            self.speaking = False

    s = Sess()
    pcm = np.full(320, 0.5, np.float32)
    mk = lambda t: [AudioFrameData(pcm, t, {"k": t}), AudioFrameData(pcm, t, {"k": t})]    # noqa: E731
    s.res_frame_queue.put((None, mk(1), 1))                      # silent: bank frame 1
    s.res_frame_queue.put((_FakePred(111), mk(0), 2))            # speaking
    s.res_frame_queue.put((_FakePred(666), mk(0), 0))            # composite fails: dropped, no audio either
    for _ in range(3):
        s.res_frame_queue.put((None, mk(2), 0))                  # custom action video: indices 0, 1, 1 (mirror)
    ev = threading.Event()
    th = threading.Thread(target=s.process_frames, args=(ev,))
    th.start()
    import time
    t0 = time.time()
    while len(s.output.video) < 5 and time.time() - t0 < 10:
        time.sleep(0.01)
    ev.set()
    th.join(timeout=5)
    assert not th.is_alive() and s.output.started and s.output.stopped
    frames = [c for c in calls if c[0] == "frame"]
    assert frames[0] == ("frame", egress.SRC_MUSETALK, 1, 0, None, False)
    assert frames[1] == ("frame", egress.SRC_MUSETALK, 2, 111, None, True)
    assert [f[4] for f in frames[2:]] == [10, 11, 11] and all(f[1] == egress.SRC_HOST for f in frames[2:])
    assert s.custom_index[2] == 3 and s.speaking is False
    assert len(s.output.video) == 5 and len(s.output.audio) == 10
    assert s.output.audio[0][0] == np.int16 and s.output.audio[0][2] == {"k": 1}
    assert calls[-1] == ("close", 7)


def test_host_i420_twin_matches_the_oracle_and_keeps_one_frame_type():
    """The odd-size custom-clip fallback of DeviceEgressMixin.process_frames converts on the host when the stream is I420
    (one frame type per stream): same bytes as the oracle's swscale restatement, both chroma sitings; odd sizes are cropped."""
    from livetalking_amd.egress import I420Frame, host_bgr_to_i420
    from oracle import egress_oracle
    rng = np.random.default_rng(8)
    img = rng.integers(0, 256, (38, 54, 3), dtype=np.uint8)
    for chroma in (1, 0):
        got = host_bgr_to_i420(img, chroma)
        assert isinstance(got, I420Frame) and (got.height, got.width) == (38, 54)
        assert np.array_equal(np.asarray(got), egress_oracle.bgr_to_i420(img, chroma))
    odd = host_bgr_to_i420(img[:37, :53])
    assert (odd.height, odd.width) == (36, 52) and np.array_equal(np.asarray(odd), egress_oracle.bgr_to_i420(img[:36, :52]))
