#!/usr/bin/env python3
"""The reference's OWN render loop, headless.  TEST INFRASTRUCTURE ONLY.

Runs the unmodified `BaseAvatar.render / inference / process_frames` threads (avatars/base_avatar.py:326-501), the
unmodified `registry.create("avatar", "wav2lip", opt=..., model=..., avatar=...)` factory (registry.py:35-50,
app.py:99), `EdgeTTS` thread and `WebRTCOutput` (streamout/webrtc.py:14-43) from a LiveTalking checkout, with a fake
player in place of `HumanPlayer` (server/webrtc.py:186-198), in one of three arrangements:

  reference    the reference's own avatars.wav2lip_avatar.LipReal + MelASR + a torch model (its Wav2Lip, or TinyLip)
  plugin-fake  THIS repo's plugin modules overlaid on the module names app.py imports (what scripts/run_amd.py does),
               with tests/fake_engine.FakeEngine behind them (CPU; control flow + glue, no HIP)
  plugin-gpu   the same overlay with the real HIP engine (needs a GPU and the checkout)

    python -m oracle.ref_loop --mode reference --ref /root/reference --net wav2lip --steps 2 --out /tmp/x.npz

Nothing from the reference is copied: it is imported from where it lies.  Third-party modules the checkout imports
and this image lacks (cv2, av, resampy, soundfile, edge_tts, librosa) are stubbed the way the reference's own test
stubs (tests/test_asr_server.py:29-72); the three arithmetic leaves (librosa.stft, librosa.filters.mel, cv2.resize)
are the restatements of oracle/mel_oracle.py and oracle/paste_oracle.py.
"""
from __future__ import annotations

import argparse
import importlib
import importlib.machinery
import logging
import os
import sys
import tempfile
import threading
import time
import types
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install_stubs():
    """Stub modules for the third-party imports of the reference's plugin layer."""
    import transformers.audio_utils  # noqa: F401  (import before stubbing: find_spec on a stub raises)
    from oracle import mel_oracle, paste_oracle

    def cv2_resize(src, dsize, *a, **k):
        return paste_oracle.resize_linear_u8(np.ascontiguousarray(src), dsize)

    _stub("cv2", resize=cv2_resize, putText=lambda *a, **k: None, FONT_HERSHEY_SIMPLEX=0, imread=lambda p: None)
    _stub("av", AudioFrame=object, VideoFrame=object)
    _stub("resampy")
    _stub("soundfile")
    _stub("edge_tts")
    filters = _stub("librosa.filters",
                    mel=lambda sr, n_fft, n_mels, fmin, fmax: mel_oracle.mel_filterbank(sr, n_fft, n_mels, fmin, fmax))
    _stub("librosa", stft=lambda y, n_fft, hop_length, win_length: mel_oracle.stft(y, n_fft, hop_length, win_length),
          filters=filters)


def enter_reference(ref_root: str):
    """Make the checkout importable and move to a scratch CWD (utils/logger.py:7 creates livetalking.log there)."""
    install_stubs()
    if ref_root not in sys.path:
        sys.path.insert(0, ref_root)
    os.chdir(tempfile.mkdtemp(prefix="ltk_refloop_"))


class FakePlayer:
    """What HumanPlayer offers the output plugin (server/webrtc.py:186-198): push_video / push_audio / buffer size."""

    def __init__(self):
        self.video = []          # copies: the silent path hands out bank frames without a copy (base_avatar.py:417)
        self.audio = []
        self.events = []
        self.order = []          # 'v' / 'a' in arrival order
        self.bad = []            # frames a real player could not have taken
        self._lock = threading.Lock()

    def push_video(self, frame):
        with self._lock:
            # what VideoFrame.from_ndarray(frame, "bgr24") needs (server/webrtc.py:190-193)
            if not (isinstance(frame, np.ndarray) and frame.dtype == np.uint8 and frame.ndim == 3 and frame.shape[2] == 3
                    and frame.flags["C_CONTIGUOUS"]):
                self.bad.append((len(self.video), type(frame).__name__, getattr(frame, "dtype", None), getattr(frame, "shape", None)))
            self.video.append(np.array(frame, copy=True))
            self.order.append("v")

    def push_audio(self, frame, eventpoint=None):
        with self._lock:
            self.audio.append(np.asarray(frame).copy())
            self.events.append(eventpoint)
            self.order.append("a")

    def get_buffer_size(self):
        return 0                 # free-running (saturating) mode, SURVEY App. D.6

    def notify(self, ev):
        pass


def tiny_lip():
    """A cheap stand-in network with the Wav2Lip.forward(mel, img) signature (wav2lip_v2.py:123): lets the CPU tests
    run the full loop in a second.  The same function backs tests/fake_engine.FakeEngine, so the reference LipReal and
    the overlaid plugin must deliver identical frames."""
    import torch

    class TinyLip(torch.nn.Module):
        def forward(self, mel, img):
            b = mel.shape[0]
            m = mel.reshape(b, -1).mean(dim=1).view(b, 1, 1, 1)
            ref = img[:, 3:6]
            up = img[:, 0:3]
            shift = torch.roll(ref, shifts=(3, -5), dims=(2, 3))
            return torch.sigmoid(2.0 * ref - 1.0 + 0.25 * m + 0.5 * up - 0.3 * shift)

        def eval(self):
            return self

    return TinyLip()


def make_opt(batch_size=16, sessionid="s0", **extra):
    """All the fields the path reads (base_avatar.py:65-68,85,103-105,117-120,199; base_asr.py:34-44)."""
    ns = argparse.Namespace(fps=25, batch_size=batch_size, l=10, r=10, sessionid=sessionid, tts="edgetts", transport="webrtc",
                            customopt=[], REF_FILE="", avatar_id="synthetic")
    for k, v in extra.items():
        setattr(ns, k, v)
    return ns


class FpsLog(logging.Handler):
    """Collects the '------actual avg infer fps' lines of base_avatar.py:364-373."""

    def __init__(self):
        super().__init__(level=logging.INFO)
        self.lines = []

    def emit(self, record):
        msg = record.getMessage()
        if "actual avg infer fps" in msg:
            self.lines.append(msg)


def run_session(session, audio: np.ndarray, n_steps: int, batch_size: int, timeout_s: float = 300.0, tail_steps: int = 0):
    """Feed `n_steps` steps worth of 20-ms speech chunks (all queued up front, so no chunk is replaced by timeout
    silence), start the reference's render thread, wait until n_steps*B (+ tail_steps*B silent) frames arrived, stop."""
    player = FakePlayer()
    session.output._player = player                     # what HumanPlayer.__init__ does (server/webrtc.py:186-188)
    # asr.warm_up() (base_asr.py:76-82) ran in the session constructor on an empty queue: its l+r chunks were timeout
    # silence.  Speech for n_steps steps plus the r look-ahead chunks the last step's windows reach into.
    n_chunks = n_steps * 2 * batch_size
    assert len(audio) >= n_chunks * 320
    for c in range(n_chunks):
        session.put_audio_frame(audio[c * 320:(c + 1) * 320].astype(np.float32), {"chunk": c})
    quit_event = threading.Event()
    t = threading.Thread(target=session.render, args=(quit_event,), name="render")
    t.start()
    want = (n_steps + tail_steps) * batch_size
    t0 = time.time()
    while len(player.video) < want and time.time() - t0 < timeout_s and t.is_alive():
        time.sleep(0.01)
    quit_event.set()
    t.join(timeout=30)
    def others():
        return [th.name for th in threading.enumerate() if th is not threading.current_thread() and th.is_alive()
                and th.name != "ltk-coalesce" and not th.daemon]
    t1 = time.time()
    while others() and time.time() - t1 < 5.0:      # the TTS thread polls its queue with a 1 s timeout (tts/base_tts.py:45)
        time.sleep(0.05)
    alive = others()
    return player, {"render_joined": not t.is_alive(), "leftover_threads": alive, "wall_s": time.time() - t0}


def summarize(player: FakePlayer, coords, n_speech_frames: int, sub: int = 4):
    """Compact, comparable record of what a session delivered."""
    v = player.video
    out = {"n_video": len(v), "n_audio": len(player.audio), "n_bad_frames": len(player.bad),
           "order": "".join(player.order[: 3 * n_speech_frames])}
    crc_full, outside_crc, box_sub = [], [], []
    for i, f in enumerate(v[:n_speech_frames]):
        crc_full.append(zlib.crc32(f.tobytes()))
        if coords is not None:
            y1, y2, x1, x2 = coords[i]
            g = f.copy()
            g[y1:y2, x1:x2] = 0
            outside_crc.append(zlib.crc32(g.tobytes()))
            box_sub.append(np.ascontiguousarray(f[y1:y2:sub, x1:x2:sub][:48, :48]))
    out["crc_full"] = np.asarray(crc_full, dtype=np.uint32)
    out["crc_outside_box"] = np.asarray(outside_crc, dtype=np.uint32)
    if box_sub:
        hh = min(b.shape[0] for b in box_sub); ww = min(b.shape[1] for b in box_sub)
        out["box_sub"] = np.stack([b[:hh, :ww] for b in box_sub])
    out["audio_crc"] = np.asarray([zlib.crc32(a.tobytes()) for a in player.audio[: 2 * n_speech_frames]], dtype=np.uint32)
    out["audio_chunk_ids"] = np.asarray([(e or {}).get("chunk", -1) if isinstance(e, dict) else -1
                                         for e in player.events[: 2 * n_speech_frames]], dtype=np.int32)
    return out


def overlay_plugin():
    """Resolve the module names app.py imports (app.py:128-137) to this repo's plugin modules: the in-process form of
    scripts/run_amd.py.  Must run after enter_reference() so that hostshim binds to the reference's own base classes."""
    import livetalking_amd.hostshim as shim
    assert shim.USING_REFERENCE_HOST, "hostshim did not bind to the reference's base classes"
    import livetalking_amd.avatars.audio_features.mel as mel
    import livetalking_amd.avatars.wav2lip_avatar as w2l
    sys.modules["avatars.wav2lip_avatar"] = w2l
    sys.modules["avatars.audio_features.mel"] = mel
    return w2l


def build_session(mode: str, ref_root: str, net: str, batch_size: int, avatar, sd_np=None, extra_opt=None):
    """Returns (session, model_handle).  `avatar` = (frames, faces, coords) as load_avatar returns it."""
    import torch
    enter_reference(ref_root)
    import registry
    opt = make_opt(batch_size=batch_size, **(extra_opt or {}))
    if mode == "reference":
        plugin = importlib.import_module("avatars.wav2lip_avatar")      # executes @register("avatar", "wav2lip")
        assert plugin.__file__.startswith(ref_root)
        if net == "tiny":
            model = tiny_lip()
        else:
            from avatars.wav2lip.models import Wav2Lip
            model = Wav2Lip().eval()
            model.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
    else:
        plugin = overlay_plugin()
        if mode == "plugin-fake":
            sys.path.insert(0, os.path.join(REPO, "tests"))
            from fake_engine import FakeEngine
            model = plugin.Wav2LipModel(FakeEngine(net=net, sd_np=sd_np))
        else:
            model = plugin.load_model(None, state_dict=sd_np, max_frames=max(16, batch_size))
            plugin.warm_up(batch_size, model, 256)
    cls = registry._REGISTRY["avatar"]["wav2lip"]
    assert cls.__module__ == plugin.__name__ or cls is plugin.LipReal, (cls, plugin)
    session = registry.create("avatar", "wav2lip", opt=opt, model=model, avatar=avatar)     # app.py:99
    return session, model


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", choices=("reference", "plugin-fake", "plugin-gpu"), required=True)
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--net", choices=("tiny", "wav2lip"), default="tiny")
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--tail-steps", type=int, default=0, help="also wait for this many all-silent steps after the speech")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--frames", type=int, default=5, help="avatar bank length (shorter than the run: exercises the ping-pong)")
    ap.add_argument("--egress", default="", help="plugin modes: opt.egress (bgr24 / i420) -> DeviceEgressMixin.process_frames")
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    args.out = os.path.abspath(args.out)        # enter_reference() moves to a scratch CWD

    import synth_inputs as synth
    sd_np = synth.wav2lip_state_dict(1234) if args.net == "wav2lip" else None
    fr, fa, co = synth.wav2lip_avatar(n_frames=args.frames, full_hw=(360, 640), box=160, seed=0)
    avatar = ([np.ascontiguousarray(f) for f in fr], [np.ascontiguousarray(f) for f in fa], co)   # as cv2.imread returns them
    audio = synth.synthetic_audio(4.0 + 0.64 * args.steps)
    extra = {"egress": args.egress} if args.egress else None
    session, model = build_session(args.mode, os.path.abspath(args.ref), args.net, args.batch, avatar, sd_np, extra)

    from utils.logger import logger as ref_logger          # the reference's logger (utils/logger.py)
    fps = FpsLog()
    ref_logger.addHandler(fps)
    player, info = run_session(session, audio, args.steps, args.batch, tail_steps=args.tail_steps)
    from utils.image import mirror_index
    n_speech = args.steps * args.batch
    coords = [avatar[2][mirror_index(len(avatar[0]), i)] for i in range(n_speech)]
    rec = summarize(player, coords, n_speech)
    rec.update(mode=args.mode, net=args.net, steps=args.steps, batch=args.batch, bank_frames=args.frames,
               render_joined=info["render_joined"], leftover_threads=np.asarray(info["leftover_threads"]),
               wall_s=info["wall_s"], fps_log=np.asarray(fps.lines))
    np.savez_compressed(args.out, **rec)
    print(f"{args.mode}/{args.net}: {rec['n_video']} video + {rec['n_audio']} audio frames in {info['wall_s']:.1f} s, "
          f"fps lines {len(fps.lines)}, joined {info['render_joined']}, leftover {info['leftover_threads']}")
    if hasattr(model, "engine") and hasattr(model.engine, "close"):
        model.engine.close()
    os._exit(0)      # the TTS thread of the reference polls its queue with a 1 s timeout; do not wait for it


if __name__ == "__main__":
    main()
