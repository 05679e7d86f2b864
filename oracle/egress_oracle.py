"""Oracle: frame egress (transition blend, watermark, BGR24 -> I420), numpy integer / float32 arithmetic on CPU.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates
  avatars/base_avatar.py:419-426, 436-445   cv2.addWeighted(last, 1-alpha, frame, alpha, 0) transition blend
  avatars/base_avatar.py:449                 cv2.putText(frame, "LiveTalking", (10,20), HERSHEY_SIMPLEX, 0.3, (128,)*3, 1)
  server/webrtc.py:190-193                   VideoFrame.from_ndarray(frame, "bgr24"); aiortc's encoders then call
                                             frame.reformat(format="yuv420p") = libswscale bgr24 -> yuv420p
  avatars/base_avatar.py:402-447             the transition clock / cache protocol (TransitionState below)

Third-party leaves (none of them vendored in /root/reference or installed here):
  * OpenCV `addWeighted` on CV_8U: `saturate_cast<uchar>(cvRound(src1*alpha + src2*beta + gamma))` evaluated in
    float32 (modules/core/src/arithm.simd.hpp, op_add_weighted: the 8-bit path converts to float, multiplies by the
    float-cast weights, adds, rounds half to even).
  * OpenCV `putText` with thickness 1 / LINE_8 sets every pixel the Hershey strokes cover to the colour.  The glyph
    table is OpenCV data, so the host rasterises the text with cv2 itself (livetalking_amd/egress.py:watermark_mask) and
    this oracle, like the kernel, takes the coverage bitmap as an input.
  * libswscale bgr24 -> yuv420p: BT.601 limited range with swscale's 15-bit integer matrix (libswscale/input.c,
    rgb2rgb_template.c; RGB2YUV_SHIFT = 15)
        RY,GY,BY = 8414,16519,3208;  RU,GU,BU = -4865,-9528,14392;  RV,GV,BV = 14392,-12061,-2332,
        Y = ((RY*r + GY*g + BY*b + 2^14) >> 15) + 16,  U = ((RU*r + GU*g + BU*b + 2^14) >> 15) + 128,  V likewise
    (round to nearest, as the generic input stage does; the unscaled shortcut ff_rgb24toyv12_c of some builds
    truncates instead and is at most 1 LSB lower).  Chroma is taken from the top-left pixel of each 2x2 quad in that
    shortcut's C routine (`chroma=0`) and from the 2x2 mean in the generic / SIMD paths (`chroma=1`).

PARITY UNPINNED: neither OpenCV nor PyAV/FFmpeg exists in this container and the reference has no golden frames for
these steps; the constants are checked against the BT.601 limited-range known answers (white 235/128/128, black
16/128/128, red 81/90/240, green 145/54/34, blue 41/240/110) in tests/test_egress.py.  The HIP kernel must match THIS
statement bit-exactly.
"""
from __future__ import annotations

import numpy as np

RY, GY, BY = 8414, 16519, 3208
RU, GU, BU = -4865, -9528, 14392
RV, GV, BV = 14392, -12061, -2332


def add_weighted_u8(a: np.ndarray, alpha: float, b: np.ndarray, beta: float) -> np.ndarray:
    """cv2.addWeighted(a, alpha, b, beta, 0) for uint8 images."""
    fa = a.astype(np.float32) * np.float32(alpha)
    fb = b.astype(np.float32) * np.float32(beta)
    return np.clip(np.rint(fa + fb), 0, 255).astype(np.uint8)


def apply_watermark(frame: np.ndarray, mask: np.ndarray, x: int, y: int, color=(128, 128, 128)) -> np.ndarray:
    """In place, like cv2.putText: pixels under the non-zero mask (placed at x, y) take the colour."""
    H, W = frame.shape[:2]
    h, w = mask.shape
    ys, xs = np.nonzero(mask)
    yy, xx = ys + y, xs + x
    ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
    frame[yy[ok], xx[ok]] = np.asarray(color, dtype=np.uint8)
    return frame


def bgr_to_i420(frame: np.ndarray, chroma: int = 1) -> np.ndarray:
    """uint8 (H, W, 3) BGR -> uint8 (H*3/2, W): Y plane, then U, then V (each (H/2)*(W/2) bytes)."""
    H, W = frame.shape[:2]
    if H % 2 or W % 2:
        raise ValueError("I420 needs even dimensions")
    f = frame.astype(np.int64)
    b, g, r = f[..., 0], f[..., 1], f[..., 2]
    Y = ((RY * r + GY * g + BY * b + 16384) >> 15) + 16
    if chroma:
        q = (f[0::2, 0::2] + f[0::2, 1::2] + f[1::2, 0::2] + f[1::2, 1::2] + 2) >> 2
    else:
        q = f[0::2, 0::2]
    cb, cg, cr = q[..., 0], q[..., 1], q[..., 2]
    U = ((RU * cr + GU * cg + BU * cb + 16384) >> 15) + 128
    V = ((RV * cr + GV * cg + BV * cb + 16384) >> 15) + 128
    out = np.concatenate([Y.reshape(-1), U.reshape(-1), V.reshape(-1)]).astype(np.uint8)
    return out.reshape(H * 3 // 2, W)


class TransitionState:
    """The per-session state of process_frames' transition effect (base_avatar.py:384-447), with the clock injected."""

    def __init__(self, enable: bool, duration: float = 0.1, clock=None, t0: float = 0.0):
        self.enable, self.duration = enable, duration
        self.clock = clock
        self.last_speaking = False
        self.transition_start = clock() if clock else t0
        self.last_frame = {False: None, True: None}      # _last_silent_frame / _last_speaking_frame

    def step(self, frame: np.ndarray, speaking: bool) -> np.ndarray:
        """frame: the silent target frame or the pasted-back speaking frame; returns combine_frame (before the
        watermark)."""
        now = self.clock()
        if speaking != self.last_speaking:
            self.transition_start = now
        self.last_speaking = speaking
        if not self.enable:
            return frame
        other = self.last_frame[not speaking]
        dt = self.clock() - self.transition_start
        if dt < self.duration and other is not None:
            alpha = min(1.0, dt / self.duration)
            out = add_weighted_u8(other, 1 - alpha, frame, alpha)
        else:
            out = frame
        self.last_frame[speaking] = out.copy()
        return out


def egress_frame(frame: np.ndarray, wm=None, fmt: str = "bgr24", chroma: int = 1) -> np.ndarray:
    """watermark + format conversion of one combine_frame."""
    out = frame.copy()
    if wm is not None:
        mask, x, y, color = wm
        apply_watermark(out, mask, x, y, color)
    return bgr_to_i420(out, chroma) if fmt == "i420" else out
