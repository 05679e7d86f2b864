"""Seeded synthetic inputs for parity tests and the bench.

TEST/BENCH INFRASTRUCTURE (see oracle/__init__.py).  There are no trained
weights, avatar banks or audio clips in the reference tree, so every parity
claim is made on these seeded stand-ins:

* weights  - He-scaled conv kernels + randomised BatchNorm affine/running stats
             under the reference state_dict names
             (avatars/wav2lip/models/wav2lip_v2.py:12-91, conv.py:5-44);
* audio    - the tone+noise formula of benchmark_asr.py:44-59;
* avatar   - smooth low-pass-noise face crops / full frames and (y1,y2,x1,x2)
             boxes in the layout `load_avatar` returns
             (avatars/wav2lip_avatar.py:72-88, avatars/wav2lip/genavatar.py:130).

Generators use numpy's PCG64 (`default_rng`) only, which is bit-stable across
platforms, so the GPU box regenerates the exact tensors the golden fixtures
were made from.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np

from . import wav2lip_oracle as W


def wav2lip_state_dict(seed: int = 1234) -> Dict[str, np.ndarray]:
    """Reference-named fp32 state_dict as numpy arrays (380 tensors)."""
    rng = np.random.default_rng(seed)
    sd: Dict[str, np.ndarray] = {}
    for l in W.all_block_layers():
        kh, kw = l.k
        if l.kind == "conv":
            shape = (l.cout, l.cin, kh, kw)      # nn.Conv2d layout
            fan_in = l.cin * kh * kw
        else:
            shape = (l.cin, l.cout, kh, kw)      # nn.ConvTranspose2d layout
            # each output pixel of the s2 transposed conv sees ~k*k/s*s taps
            fan_in = l.cin * kh * kw / (l.stride[0] * l.stride[1])
        std = np.sqrt(2.0 / fan_in)
        sd[l.prefix + ".conv_block.0.weight"] = (rng.standard_normal(shape) * std).astype(np.float32)
        sd[l.prefix + ".conv_block.0.bias"] = (rng.standard_normal(l.cout) * 0.05).astype(np.float32)
        # residual layers: damp the conv branch so y = relu(bn(conv(x)) + x) stays bounded
        g_lo, g_hi = (0.3, 0.7) if l.residual else (0.7, 1.3)
        sd[l.prefix + ".conv_block.1.weight"] = rng.uniform(g_lo, g_hi, l.cout).astype(np.float32)
        sd[l.prefix + ".conv_block.1.bias"] = (rng.standard_normal(l.cout) * 0.1).astype(np.float32)
        sd[l.prefix + ".conv_block.1.running_mean"] = (rng.standard_normal(l.cout) * 0.2).astype(np.float32)
        sd[l.prefix + ".conv_block.1.running_var"] = rng.uniform(0.6, 1.6, l.cout).astype(np.float32)
        sd[l.prefix + ".conv_block.1.num_batches_tracked"] = np.asarray(1000, dtype=np.int64)
    # output head: plain conv 32->3; scaled so the sigmoid is used across its range
    sd[W.OUTPUT_HEAD_PREFIX + ".weight"] = (rng.standard_normal((3, 32, 1, 1)) * 0.04).astype(np.float32)
    sd[W.OUTPUT_HEAD_PREFIX + ".bias"] = (rng.standard_normal(3) * 0.2).astype(np.float32)
    return sd


def synthetic_audio(duration_s: float, sample_rate: int = 16000, seed: int = 42) -> np.ndarray:
    """benchmark_asr.py:44-59 (same constants, same seed by default)."""
    rng = np.random.default_rng(seed)
    t = np.linspace(0, duration_s, int(sample_rate * duration_s), dtype=np.float32)
    audio = (
        0.3 * np.sin(2 * np.pi * 200 * t)
        + 0.2 * np.sin(2 * np.pi * 500 * t)
        + 0.1 * np.sin(2 * np.pi * 1200 * t)
        + 0.15 * rng.standard_normal(len(t)).astype(np.float32)
    )
    fade = int(0.05 * sample_rate)
    audio[:fade] *= np.linspace(0, 1, fade)
    audio[-fade:] *= np.linspace(1, 0, fade)
    return audio.astype(np.float32)


def _smooth_image(rng: np.random.Generator, h: int, w: int, cells: int = 12) -> np.ndarray:
    """Low-pass noise image uint8 (h,w,3): bilinear upsample of a coarse random
    grid plus a little fine grain, so bilinear resize / PSNR are meaningful."""
    gh, gw = cells + 1, cells * w // h + 2
    coarse = rng.uniform(0, 255, (gh, gw, 3))
    ys = np.linspace(0, gh - 1.001, h)
    xs = np.linspace(0, gw - 1.001, w)
    y0 = ys.astype(np.int64)
    x0 = xs.astype(np.int64)
    fy = (ys - y0)[:, None, None]
    fx = (xs - x0)[None, :, None]
    a = coarse[y0][:, x0]
    b = coarse[y0][:, x0 + 1]
    c = coarse[y0 + 1][:, x0]
    d = coarse[y0 + 1][:, x0 + 1]
    img = (a * (1 - fy) * (1 - fx) + b * (1 - fy) * fx + c * fy * (1 - fx) + d * fy * fx)
    img += rng.normal(0, 3.0, img.shape)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def wav2lip_avatar(n_frames: int = 8, full_hw: Tuple[int, int] = (720, 1280),
                   box: int = 320, seed: int = 0
                   ) -> Tuple[List[np.ndarray], List[np.ndarray], List[Tuple[int, int, int, int]]]:
    """(frame_list_cycle, face_list_cycle, coord_list_cycle) as `load_avatar`
    returns them (avatars/wav2lip_avatar.py:72-88): BGR uint8 full frames,
    BGR uint8 256x256 face crops, (y1,y2,x1,x2) int boxes that differ per frame
    (avatars/wav2lip/genavatar.py:118-130)."""
    rng = np.random.default_rng(seed)
    H, W_ = full_hw
    frames, faces, coords = [], [], []
    cy, cx = H // 2, W_ // 2
    for _ in range(n_frames):
        frames.append(_smooth_image(rng, H, W_))
        faces.append(_smooth_image(rng, 256, 256, cells=10))
        j = rng.integers(-4, 5, 4)
        y1 = int(cy - box // 2 + j[0]); y2 = int(cy + box // 2 + j[1])
        x1 = int(cx - box // 2 + j[2]); x2 = int(cx + box // 2 + j[3])
        coords.append((y1, y2, x1, x2))
    return frames, faces, coords
