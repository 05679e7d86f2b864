"""Device-side frame egress: the reference's `BaseAvatar.process_frames`
(avatars/base_avatar.py:384-460) with everything between `res_frame_queue.get` and `output.push_video_frame` done on
the GPU in one call per frame: paste-back composite (or the silent / custom frame), the speaking<->silent transition
`cv2.addWeighted`, the "LiveTalking" watermark, and optionally BGR24 -> I420 so that the host copies 1.5 instead of
3 bytes per pixel and the encoder skips its swscale pass (server/webrtc.py:190-193).

Opt-in: `opt.egress` in {"bgr24", "i420"} makes the plugin classes use `process_frames` below instead of the base
class's loop; unset, the reference's own loop runs unchanged and calls `paste_back_frame`.
"""
from __future__ import annotations

import queue
import time

import numpy as np

FMT_BGR24, FMT_I420 = 0, 1
SRC_WAV2LIP, SRC_MUSETALK, SRC_HOST = 0, 1, 2

WATERMARK_TEXT = "LiveTalking"          # base_avatar.py:449
WATERMARK_ORG = (10, 20)
WATERMARK_COLOR = (128, 128, 128)


def watermark_mask(H: int, W: int):
    """Coverage bitmap of the reference's cv2.putText call, rasterised by OpenCV itself so the glyphs are its own.
    Returns (mask uint8 [h][w], x, y) or None when OpenCV is not installed (tests / the GPU box)."""
    try:
        import cv2  # type: ignore
    except Exception:  # noqa: BLE001
        return None
    canvas = np.zeros((H, W), dtype=np.uint8)
    cv2.putText(canvas, WATERMARK_TEXT, WATERMARK_ORG, cv2.FONT_HERSHEY_SIMPLEX, 0.3, 255, 1)
    ys, xs = np.nonzero(canvas)
    if ys.size == 0:
        return None
    y0, y1, x0, x1 = int(ys.min()), int(ys.max()) + 1, int(xs.min()), int(xs.max()) + 1
    return np.ascontiguousarray(canvas[y0:y1, x0:x1]), x0, y0


class FrameGroup:
    """The B device predictions of ONE inference_batch call and the bank frame of each (shared by the B items the call returned):
    the process thread asks for them one by one, in order (base_avatar.py:429-433); the first request composites all B on the GPU
    and moves them to the host in one copy.  The bank indices are `mirror_index(n, index + i)`; they are worked out when the first
    consumer asks (in ITS thread): the inference thread's time between two engine calls is time the GPU idles."""
    __slots__ = ("pred", "_idx", "_span", "host", "eg_host", "eg_key", "lock")

    def __init__(self, pred, idx=None, span=None):
        import threading
        self.pred, self._idx, self._span = pred, idx, span      # span = (bank length, first index, count)
        self.host = None            # paste_back_frame path: numpy [B][H][W][3] over a pinned block
        self.eg_host, self.eg_key = None, None      # device-egress path: numpy over a pinned block of converted frames
        self.lock = threading.Lock()

    @property
    def idx(self):
        ix = self._idx
        if ix is None:
            from .hostshim import mirror_index
            n, first, count = self._span
            ix = self._idx = [mirror_index(n, first + i) for i in range(count)]
        return ix

    @staticmethod
    def attach(items, pred, idx=None, span=None):
        grp = FrameGroup(pred, idx, span)
        for i, it in enumerate(items):
            it._ltk_group, it._ltk_i = grp, i
        return grp


class I420Frame(np.ndarray):
    """uint8 [H*3/2][W] planar YUV 4:2:0 as `av.VideoFrame.from_ndarray(..., format="yuv420p")` takes it."""
    width: int
    height: int

    def planes(self):
        H, W = self.height, self.width
        flat = np.asarray(self).reshape(-1)
        y = flat[: H * W].reshape(H, W)
        u = flat[H * W: H * W + (H // 2) * (W // 2)].reshape(H // 2, W // 2)
        v = flat[H * W + (H // 2) * (W // 2):].reshape(H // 2, W // 2)
        return y, u, v


def host_bgr_to_i420(frame: np.ndarray, chroma: int = 1) -> "I420Frame":
    """Host twin of the device BGR24 -> I420 conversion (csrc/egress_kernels.hip; libswscale's 15-bit BT.601 limited-range
    matrix, round to nearest; chroma from the 2x2 mean, or from the top-left pixel when chroma == 0), for the one frame kind
    that cannot go through the engine's egress session: a custom-action clip whose size differs from the avatar's
    (base_avatar.py:410-416).  An odd trailing row / column is dropped (4:2:0 needs even dimensions)."""
    H, W = frame.shape[0] & ~1, frame.shape[1] & ~1
    f = np.ascontiguousarray(frame[:H, :W]).astype(np.int32)
    b, g, r = f[..., 0], f[..., 1], f[..., 2]
    y = ((8414 * r + 16519 * g + 3208 * b + 16384) >> 15) + 16
    q = ((f[0::2, 0::2] + f[0::2, 1::2] + f[1::2, 0::2] + f[1::2, 1::2] + 2) >> 2) if chroma else f[0::2, 0::2]
    cb, cg, cr = q[..., 0], q[..., 1], q[..., 2]
    u = ((-4865 * cr - 9528 * cg + 14392 * cb + 16384) >> 15) + 128
    v = ((14392 * cr - 12061 * cg - 2332 * cb + 16384) >> 15) + 128
    out = np.concatenate([y.reshape(-1), u.reshape(-1), v.reshape(-1)]).astype(np.uint8).reshape(H * 3 // 2, W).view(I420Frame)
    out.width, out.height = W, H
    return out


class DeviceEgress:
    """One per render session: the engine-side egress session plus the transition clock of process_frames."""

    def __init__(self, engine, H: int, W: int, source: int, avatar_id: int, fmt: str = "bgr24", enable_transition: bool = False,
                 transition_duration: float = 0.1, watermark="auto", chroma: int = 1, clock=time.time):
        if fmt not in ("bgr24", "i420"):
            raise ValueError("egress format must be 'bgr24' or 'i420'")
        if fmt == "i420" and (H % 2 or W % 2):
            raise ValueError("I420 needs even frame dimensions")
        self.engine, self.H, self.W = engine, int(H), int(W)
        self.source, self.avatar_id = int(source), int(avatar_id)
        self.fmt = FMT_I420 if fmt == "i420" else FMT_BGR24
        self.chroma = int(chroma)
        self.enable_transition = bool(enable_transition)          # base_avatar.py:384
        self.transition_duration = float(transition_duration)     # :389
        self._clock = clock
        self._last_speaking = False                               # :386
        self._transition_start = clock()                          # :387
        self._h = engine.egress_open(self.H, self.W)
        if watermark == "auto":
            watermark = watermark_mask(self.H, self.W)
        if watermark is not None:
            mask, x, y = watermark
            engine.egress_watermark(self._h, mask, x, y, WATERMARK_COLOR)

    def close(self):
        if self._h:
            self.engine.egress_close(self._h)
            self._h = 0

    def _alpha(self, speaking: bool) -> float:
        # base_avatar.py:402-406: a state change restarts the transition clock
        if speaking != self._last_speaking:
            self._transition_start = self._clock()
        self._last_speaking = speaking
        if not self.enable_transition:
            return -1.0
        dt = self._clock() - self._transition_start
        if dt < self.transition_duration:                         # :421 / :438
            return min(1.0, dt / self.transition_duration)
        return -1.0

    def _out(self):
        if self.fmt == FMT_I420:
            out = np.empty((self.H * 3 // 2, self.W), dtype=np.uint8).view(I420Frame)
            out.width, out.height = self.W, self.H
            return out
        return np.empty((self.H, self.W, 3), dtype=np.uint8)

    def speaking_frame(self, d_pred_ptr: int, idx: int) -> np.ndarray:
        """base_avatar.py:429-447: paste_back_frame + silent->speaking blend (+ watermark, format)."""
        alpha = self._alpha(True)
        return self.engine.egress_frame(self._h, self._out(), self.source, self.avatar_id, idx, d_pred_ptr, None, True, alpha,
                                        self.enable_transition, self.fmt, self.chroma)

    def speaking_frame_of(self, res_frame, idx: int) -> np.ndarray:
        """speaking_frame for an ITEM of inference_batch (a device tensor that may carry its FrameGroup): without the
        transition effect (the reference's default) the first frame of a batch converts all of them - composite, watermark,
        format - and copies them to the host at once (ltk_egress_batch); the returned frames are views of one pinned block that
        lives as long as any of them."""
        grp = getattr(res_frame, "_ltk_group", None)
        if (grp is None or self.enable_transition or not hasattr(self.engine, "egress_batch")
                or grp.idx[res_frame._ltk_i] != int(idx)):
            return self.speaking_frame(res_frame.data_ptr(), idx)
        self._alpha(True)                                   # keeps the speaking / silent state machine in step
        key = (id(self), self.fmt)
        with grp.lock:
            if grp.eg_host is None or grp.eg_key != key:
                import torch
                n = len(grp.idx)
                shape = (n, self.H * 3 // 2, self.W) if self.fmt == FMT_I420 else (n, self.H, self.W, 3)
                host = torch.empty(shape, dtype=torch.uint8, pin_memory=True)
                self.engine.egress_batch(self._h, self.source, self.avatar_id, grp.idx, grp.pred.data_ptr(), host.data_ptr(), self.fmt,
                                         self.chroma)
                grp.eg_host, grp.eg_key = host.numpy(), key
        out = grp.eg_host[res_frame._ltk_i]
        if self.fmt == FMT_I420:
            out = out.view(I420Frame)
            out.width, out.height = self.W, self.H
        return out

    def silent_frame(self, idx: int, custom_frame: np.ndarray = None) -> np.ndarray:
        """base_avatar.py:408-428: the cached full frame (or a custom-action frame) + speaking->silent blend."""
        alpha = self._alpha(False)
        if custom_frame is not None:
            custom_frame = np.ascontiguousarray(custom_frame, dtype=np.uint8)
            if custom_frame.shape != (self.H, self.W, 3):
                raise ValueError("custom frame size differs from the avatar's")
            return self.engine.egress_frame(self._h, self._out(), SRC_HOST, 0, 0, 0, custom_frame, False, alpha,
                                            self.enable_transition, self.fmt, self.chroma)
        return self.engine.egress_frame(self._h, self._out(), self.source, self.avatar_id, idx, 0, None, False, alpha,
                                        self.enable_transition, self.fmt, self.chroma)


class DeviceEgressMixin:
    """`process_frames` for LipReal / MuseReal when `opt.egress` is set: the control flow of
    avatars/base_avatar.py:384-460 (queue protocol, speaking flag, custom-action index, audio push, recording), with the
    per-frame pixel work delegated to DeviceEgress.  `_egress_source` is set by the plugin class."""

    _egress_source = SRC_WAV2LIP

    def _make_egress(self):
        h, w = self.frame_list_cycle[0].shape[:2]
        return DeviceEgress(getattr(self, "engine", None) or self.model.engine, h, w, self._egress_source, self._aid, fmt=getattr(self.opt, "egress", "bgr24"),
                            enable_transition=bool(getattr(self.opt, "enable_transition", False)))

    def process_frames(self, quit_event, *args, **kwargs):
        if not getattr(self.opt, "egress", None):
            return super().process_frames(quit_event, *args, **kwargs)
        from .hostshim import mirror_index
        eg = self._make_egress()
        output = getattr(self, "output", None)
        if output is not None:
            output.start()
        try:
            while not quit_event.is_set():
                try:
                    res_frame, audio_frames, idx = self.res_frame_queue.get(block=True, timeout=1)
                except queue.Empty:
                    continue
                if audio_frames[0].type != 0 and audio_frames[1].type != 0:      # :407 all silence
                    self.speaking = False
                    audiotype = audio_frames[0].type
                    custom = None
                    cidx = getattr(self, "custom_index", {})
                    if cidx.get(audiotype) is not None:                          # :410-414
                        cyc = self.custom_img_cycle[audiotype]
                        custom = cyc[mirror_index(len(cyc), cidx[audiotype])]
                        cidx[audiotype] += 1
                    try:
                        frame = eg.silent_frame(idx, custom)
                    except ValueError:
                        # a custom-action clip of another size than the avatar: the reference pushes it as it is
                        # (base_avatar.py:410-416, 449-452); host frame, host watermark when OpenCV is there
                        frame = np.ascontiguousarray(custom)
                        try:
                            import cv2  # type: ignore
                            cv2.putText(frame, WATERMARK_TEXT, WATERMARK_ORG, cv2.FONT_HERSHEY_SIMPLEX, 0.3, WATERMARK_COLOR, 1)
                        except Exception:  # noqa: BLE001
                            pass
                        if eg.fmt == FMT_I420:
                            # keep ONE frame type per stream: the odd-sized clip is converted on the host with the same
                            # BT.601 limited-range integer arithmetic the device kernel uses
                            frame = host_bgr_to_i420(frame, eg.chroma)
                    except Exception as e:  # noqa: BLE001 - log and drop the frame, like the speaking path below
                        import logging
                        logging.getLogger(__name__).warning("silent frame error: %s", e)
                        continue
                else:
                    self.speaking = True
                    try:
                        frame = eg.speaking_frame_of(res_frame, idx)
                    except Exception as e:  # noqa: BLE001 - base_avatar.py:432-436 logs and drops the frame
                        import logging
                        logging.getLogger(__name__).warning("paste_back_frame error: %s", e)
                        continue
                if output is not None:
                    output.push_video_frame(frame)
                if hasattr(self, "record_video_data"):
                    self.record_video_data(frame)
                for af in audio_frames:                                          # :455-460
                    pcm = (af.data * 32767).astype(np.int16)
                    if output is not None:
                        output.push_audio_frame(pcm, af.userdata)
                    if hasattr(self, "record_audio_data"):
                        self.record_audio_data(pcm)
        finally:
            eg.close()
            if output is not None and hasattr(output, "stop"):
                output.stop()
