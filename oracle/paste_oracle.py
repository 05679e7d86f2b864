"""Oracle: mouth-region paste-back composite, numpy integer arithmetic on CPU.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates
  avatars/wav2lip_avatar.py:141-147  LipReal.paste_back_frame
  utils/image.py:26-32               mirror_index

Third-party leaf: `cv2.resize(src_u8, (w,h))` with the default INTER_LINEAR
(opencv-python-headless is NOT vendored in the reference, unpinned in
requirements.txt:35, and not installed here).  `resize_linear_u8` restates
OpenCV's published 8-bit bilinear path (modules/imgproc/src/resize.cpp):
half-pixel centres computed in float32, 11-bit fixed-point coefficients
(INTER_RESIZE_COEF_BITS), horizontal pass into int32, vertical pass
`((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2`, no antialiasing when
shrinking, and the exact-2x-shrink case redirected to the 2x2 box average
(INTER_AREA fast path).

PARITY UNPINNED for this leaf: no OpenCV build is available in this container
and the reference holds no golden frames, so the restatement could not be
checked against real cv2 output; the reference-side tolerance is therefore
stated as +-1 LSB.  The HIP kernel is required to match THIS
statement bit-exactly.
"""
from __future__ import annotations

import numpy as np

COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS


def mirror_index(size: int, index: int) -> int:
    """utils/image.py:26-32."""
    turn = index // size
    res = index % size
    if turn % 2 == 0:
        return res
    return size - res - 1


def _axis_tables(dst: int, src: int):
    """Per-output-coordinate source index and (1-f, f) fixed-point weights."""
    scale = np.float64(src) / np.float64(dst)          # = 1/inv_scale
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)   # float32, as in resize.cpp
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    return s, f


def _coefs(f: np.ndarray):
    c0 = (np.float32(1.0) - f).astype(np.float32)
    # saturate_cast<short>(float) rounds half to even (cvRound)
    a0 = np.rint(c0 * np.float32(COEF_SCALE)).astype(np.int64)
    a1 = np.rint(f * np.float32(COEF_SCALE)).astype(np.int64)
    return a0, a1


def resize_linear_u8(src: np.ndarray, dsize_wh) -> np.ndarray:
    """cv2.resize(src, (w, h)) for uint8 (H,W,C) input, INTER_LINEAR."""
    assert src.dtype == np.uint8 and src.ndim == 3
    dw, dh = int(dsize_wh[0]), int(dsize_wh[1])
    sh, sw, _ = src.shape
    if dw == sw and dh == sh:
        return src.copy()
    if sw == 2 * dw and sh == 2 * dh:
        s = src.astype(np.int32)
        return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)

    sx, fx = _axis_tables(dw, sw)
    # horizontal border handling (resize.cpp: sx<0 -> fx=0,sx=0 ; sx>=w-1 -> fx=0,sx=w-1)
    lo = sx < 0
    fx = np.where(lo, np.float32(0), fx); sx = np.where(lo, 0, sx)
    hi = sx >= sw - 1
    fx = np.where(hi, np.float32(0), fx); sx = np.where(hi, sw - 1, sx)
    ax0, ax1 = _coefs(fx.astype(np.float32))
    sx1 = np.minimum(sx + 1, sw - 1)

    sy, fy = _axis_tables(dh, sh)
    by0, by1 = _coefs(fy)
    sy0 = np.clip(sy, 0, sh - 1)
    sy1 = np.clip(sy + 1, 0, sh - 1)

    s = src.astype(np.int64)
    # horizontal pass: int rows D[x] = S[sx]*a0 + S[sx+1]*a1
    rows = s[:, sx, :] * ax0[None, :, None] + s[:, sx1, :] * ax1[None, :, None]
    S0 = rows[sy0]
    S1 = rows[sy1]
    out = (((by0[:, None, None] * (S0 >> 4)) >> 16) + ((by1[:, None, None] * (S1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def paste_back_frame(pred_frame: np.ndarray, full_frame: np.ndarray, bbox) -> np.ndarray:
    """wav2lip_avatar.py:141-147.  pred_frame float (256,256,3) in [0,255];
    astype(uint8) truncates toward zero."""
    y1, y2, x1, x2 = (int(v) for v in bbox)
    combine = full_frame.copy()
    res = resize_linear_u8(pred_frame.astype(np.uint8), (x2 - x1, y2 - y1))
    combine[y1:y2, x1:x2] = res
    return combine


# ---------------------------------------------------------------------------------------------
# MuseTalk paste-back: avatars/musetalk_avatar.py:154-164 + avatars/musetalk/myutil.py:4-25
# ---------------------------------------------------------------------------------------------
def blend_linear_u8(src1: np.ndarray, src2: np.ndarray, w1: np.ndarray, w2: np.ndarray) -> np.ndarray:
    """cv2.blendLinear for 8UC3 restated from OpenCV's published algorithm (modules/imgproc/src/blend.cpp):
    dst = saturate_cast<uchar>((src1*w1 + src2*w2) / (w1 + w2 + 1e-5f)), float32 arithmetic, cvRound
    (round half to even).  PARITY UNPINNED (no OpenCV here)."""
    f = np.float32
    den = (w1.astype(f) + w2.astype(f)) + f(1e-5)
    num = src1.astype(f) * w1.astype(f)[..., None] + src2.astype(f) * w2.astype(f)[..., None]
    q = num / den[..., None]
    return np.clip(np.rint(q), 0, 255).astype(np.uint8)


def cvt_bgr2gray_u8(img_bgr: np.ndarray) -> np.ndarray:
    """cv2.cvtColor(img, cv2.COLOR_BGR2GRAY) for 8UC3, restated from OpenCV's published 8-bit path
    (modules/imgproc/src/color_rgb.simd.hpp, RGB2Gray<uchar>): 14-bit fixed point,
    gray = (B*1868 + G*9617 + R*4899 + 8192) >> 14 (coefficients sum to 16384, so B = G = R = v gives v: the saved
    MuseTalk masks are grey, which is why paste_blend_frame below reads channel 0).  PARITY UNPINNED (no OpenCV here)."""
    a = np.asarray(img_bgr).astype(np.int64)
    return ((a[..., 0] * 1868 + a[..., 1] * 9617 + a[..., 2] * 4899 + 8192) >> 14).astype(np.uint8)


def paste_blend_frame(pred_frame: np.ndarray, ori_frame: np.ndarray, bbox, mask_bgr: np.ndarray, crop_box) -> np.ndarray:
    """MuseReal.paste_back_frame on explicit inputs (musetalk_avatar.py:154-164, myutil.py:4-25)."""
    x1, y1, x2, y2 = [int(v) for v in bbox]
    body = ori_frame.copy()
    res = resize_linear_u8(np.ascontiguousarray(pred_frame).astype(np.uint8), (x2 - x1, y2 - y1))
    x_s, y_s, x_e, y_e = [int(v) for v in crop_box]
    face_large = body[y_s:y_e, x_s:x_e].copy()
    face_large[y1 - y_s:y2 - y_s, x1 - x_s:x2 - x_s] = res
    gray = mask_bgr[:, :, 0]            # cvtColor(BGR2GRAY) of a B=G=R image is the channel itself
    mask_image = (gray / 255).astype(np.float32)
    body[y_s:y_e, x_s:x_e] = blend_linear_u8(face_large, body[y_s:y_e, x_s:x_e], mask_image, 1 - mask_image)
    return body
