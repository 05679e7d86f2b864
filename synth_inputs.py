"""Seeded synthetic inputs for parity tests and the bench.

Input generators only - no reference arithmetic lives here, and nothing under
oracle/ is imported.  There are no trained
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


# (state_dict prefix, kind, cin, cout, k, stride_product, residual) for the 54 Conv2d/Conv2dTranspose
# blocks of avatars/wav2lip/models/wav2lip_v2.py:12-89, in state_dict order: only what the
# generator needs (shapes + fan-in).  The engine's own table lives in csrc/engine.hip.
def _layers():
    L = []
    def c(p, cin, cout, k, s=1, res=False): L.append((p, "conv", cin, cout, k, s, res))
    def t(p, cin, cout, k, s): L.append((p, "convT", cin, cout, k, s, False))
    a = "audio_encoder."
    c(a + "0", 1, 32, 3); c(a + "1", 32, 32, 3, 1, True); c(a + "2", 32, 32, 3, 1, True)
    c(a + "3", 32, 64, 3, 3); c(a + "4", 64, 64, 3, 1, True); c(a + "5", 64, 64, 3, 1, True)
    c(a + "6", 64, 128, 3, 9); c(a + "7", 128, 128, 3, 1, True); c(a + "8", 128, 128, 3, 1, True)
    c(a + "9", 128, 256, 3, 6); c(a + "10", 256, 256, 3, 1, True); c(a + "11", 256, 512, 3); c(a + "12", 512, 512, 1)
    e = "face_encoder_blocks."
    c(e + "0.0", 6, 16, 7)
    for b, (ci, co, n) in enumerate([(16, 32, 2), (32, 64, 3), (64, 128, 2), (128, 256, 2), (256, 512, 1), (512, 512, 1)], start=1):
        c(e + f"{b}.0", ci, co, 3, 4)
        for j in range(1, n + 1): c(e + f"{b}.{j}", co, co, 3, 1, True)
    c(e + "7.0", 512, 512, 4); c(e + "7.1", 512, 512, 1)
    d = "face_decoder_blocks."
    c(d + "0.0", 512, 512, 1)
    t(d + "1.0", 1024, 512, 4, 1); c(d + "1.1", 512, 512, 3, 1, True)
    for b, (ci, co, n) in enumerate([(1024, 512, 1), (1024, 512, 2), (768, 384, 2), (512, 256, 2), (320, 128, 2), (160, 64, 2)], start=2):
        t(d + f"{b}.0", ci, co, 3, 4)
        for j in range(1, n + 1): c(d + f"{b}.{j}", co, co, 3, 1, True)
    c("output_block.0", 80, 32, 3)
    return L

OUTPUT_HEAD_PREFIX = "output_block.1"


def wav2lip_state_dict(seed: int = 1234, gain: float = 1.0) -> Dict[str, np.ndarray]:
    """Reference-named fp32 state_dict as numpy arrays (380 tensors).

    `gain` (parity-stress family, tests/test_parity_stress_gpu.py): the BatchNorm affine of the two input layers
    (face_encoder_blocks.0.0, audio_encoder.0) is multiplied by it, so every activation behind them grows about
    linearly with it (peak |activation| of the fp32 forward: ~70 at gain 1, ~1.2e3 at 16, ~1.9e4 at 256, past the
    fp16 limit 65 504 from ~1024 on), and the output head's weights are divided by it so that the sigmoid keeps
    working across its range.  gain = 1 is the draw every other test uses."""
    rng = np.random.default_rng(seed)
    sd: Dict[str, np.ndarray] = {}
    for prefix, kind, cin, cout, k, sprod, residual in _layers():
        if kind == "conv":
            shape = (cout, cin, k, k)            # nn.Conv2d layout
            fan_in = cin * k * k
        else:
            shape = (cin, cout, k, k)            # nn.ConvTranspose2d layout
            # each output pixel of the s2 transposed conv sees ~k*k/s*s taps
            fan_in = cin * k * k / sprod
        std = np.sqrt(2.0 / fan_in)
        sd[prefix + ".conv_block.0.weight"] = (rng.standard_normal(shape) * std).astype(np.float32)
        sd[prefix + ".conv_block.0.bias"] = (rng.standard_normal(cout) * 0.05).astype(np.float32)
        # residual layers: damp the conv branch so y = relu(bn(conv(x)) + x) stays bounded
        g_lo, g_hi = (0.3, 0.7) if residual else (0.7, 1.3)
        sd[prefix + ".conv_block.1.weight"] = rng.uniform(g_lo, g_hi, cout).astype(np.float32)
        sd[prefix + ".conv_block.1.bias"] = (rng.standard_normal(cout) * 0.1).astype(np.float32)
        sd[prefix + ".conv_block.1.running_mean"] = (rng.standard_normal(cout) * 0.2).astype(np.float32)
        sd[prefix + ".conv_block.1.running_var"] = rng.uniform(0.6, 1.6, cout).astype(np.float32)
        sd[prefix + ".conv_block.1.num_batches_tracked"] = np.asarray(1000, dtype=np.int64)
    # output head: plain conv 32->3; scaled so the sigmoid is used across its range
    sd[OUTPUT_HEAD_PREFIX + ".weight"] = (rng.standard_normal((3, 32, 1, 1)) * 0.04).astype(np.float32)
    sd[OUTPUT_HEAD_PREFIX + ".bias"] = (rng.standard_normal(3) * 0.2).astype(np.float32)
    if gain != 1.0:
        g = np.float32(gain)
        for p in ("face_encoder_blocks.0.0", "audio_encoder.0"):
            sd[p + ".conv_block.1.weight"] = sd[p + ".conv_block.1.weight"] * g
            sd[p + ".conv_block.1.bias"] = sd[p + ".conv_block.1.bias"] * g
        sd[OUTPUT_HEAD_PREFIX + ".weight"] = sd[OUTPUT_HEAD_PREFIX + ".weight"] / g
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


def _lerp_axis(a: np.ndarray, n_out: int, axis: int) -> np.ndarray:
    """Linear interpolation of fp32 `a` along `axis` at the sample positions `_smooth_image` uses: elementwise IEEE
    products and sums only (no BLAS), so the bytes are the same on every host."""
    n_in = a.shape[axis]
    pos = np.linspace(0, n_in - 1.001, n_out)
    i0 = pos.astype(np.int64)
    shape = [1] * a.ndim
    shape[axis] = n_out
    f = (pos - i0).astype(np.float32).reshape(shape)
    return np.take(a, i0, axis=axis) * (np.float32(1) - f) + np.take(a, i0 + 1, axis=axis) * f


def _smooth_image_fast(rng: np.random.Generator, h: int, w: int, grain: np.ndarray, cells: int = 12) -> np.ndarray:
    """`_smooth_image` for LARGE banks (the 250-frame 720p bench bank): the same coarse-grid bilinear
    field in fp32, separable, plus a fixed fine-grain field rolled by a per-frame offset (~10x faster per 720p frame).  A
    different pixel stream than `_smooth_image`: fixtures name the generator they used."""
    gh, gw = cells + 1, cells * w // h + 2
    coarse = rng.uniform(0, 255, (gh, gw, 3)).astype(np.float32)
    img = _lerp_axis(_lerp_axis(coarse, h, 0), w, 1)
    dy, dx = (int(v) for v in rng.integers(0, 64, 2))
    img += grain[dy:dy + h, dx:dx + w]
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def wav2lip_bank(n_frames: int = 250, full_hw: Tuple[int, int] = (720, 1280), box: int = 320, seed: int = 0
                 ) -> Tuple[List[np.ndarray], List[np.ndarray], List[Tuple[int, int, int, int]]]:
    """The SURVEY.md 8d bench bank: `n_frames` smooth full frames + 256^2 face crops + jittered (y1,y2,x1,x2) boxes of
    about box^2 pixels (box 320: the paste-back upscales the 256^2 prediction, box 200: it shrinks it), in the layout
    `load_avatar` returns (avatars/wav2lip_avatar.py:72-88).  Same contract as `wav2lip_avatar`, ~15x faster per frame."""
    rng = np.random.default_rng([seed, n_frames, box])
    H, W_ = full_hw
    grain = rng.normal(0, 3.0, (H + 64, W_ + 64, 3)).astype(np.float32)
    frames, faces, coords = [], [], []
    cy, cx = H // 2, W_ // 2
    for _ in range(n_frames):
        frames.append(_smooth_image_fast(rng, H, W_, grain))
        faces.append(_smooth_image_fast(rng, 256, 256, grain, cells=10))
        j = rng.integers(-4, 5, 4)
        coords.append((int(cy - box // 2 + j[0]), int(cy + box // 2 + j[1]), int(cx - box // 2 + j[2]), int(cx + box // 2 + j[3])))
    return frames, faces, coords


# ---------------------------------------------------------------------------------------------------------
# MuseTalk: seeded state dicts under diffusers' key names (UNet2DConditionModel with the MuseTalk-1.5 config,
# AutoencoderKL decoder of sd-vae-ft-mse).  No checkpoint or config exists in the reference tree
# (avatars/musetalk/utils/utils.py:16-18 loads models/musetalkV15/{unet.pth,musetalk.json} at run time).
# ---------------------------------------------------------------------------------------------------------
UNET_CH = (320, 640, 1280, 1280)
UNET_CTX = 384
VAE_CH = (128, 256, 512, 512)


class _Shape:
    """Stand-in array for `shapes_only=True`: carries a shape through the generators' arithmetic without allocating."""

    def __init__(self, shape):
        self.shape = tuple(shape) if isinstance(shape, (tuple, list)) else (int(shape),)

    def __mul__(self, other):
        return self

    def astype(self, dtype):
        return self


class _ShapeRng:
    def standard_normal(self, shape, dtype=None):
        return _Shape(shape)

    def uniform(self, lo, hi, shape):
        return _Shape(shape)


def _finish(sd, shapes_only):
    return {k: tuple(v.shape) for k, v in sd.items()} if shapes_only else sd


def _w(rng, shape, fan_in, gain=1.0):
    return (rng.standard_normal(shape, dtype=np.float32) * np.float32(gain / np.sqrt(fan_in))).astype(np.float32)


def _norm(rng, sd, p, c):
    sd[p + ".weight"] = rng.uniform(0.8, 1.2, c).astype(np.float32)
    sd[p + ".bias"] = (rng.standard_normal(c) * 0.05).astype(np.float32)


def _conv(rng, sd, p, cin, cout, k, gain=1.0):
    sd[p + ".weight"] = _w(rng, (cout, cin, k, k), cin * k * k, gain)
    sd[p + ".bias"] = (rng.standard_normal(cout) * 0.02).astype(np.float32)


def _lin(rng, sd, p, cin, cout, bias=True, gain=1.0):
    sd[p + ".weight"] = _w(rng, (cout, cin), cin, gain)
    if bias:
        sd[p + ".bias"] = (rng.standard_normal(cout) * 0.02).astype(np.float32)


def _resnet(rng, sd, p, cin, cout, temb):
    _norm(rng, sd, p + ".norm1", cin)
    _conv(rng, sd, p + ".conv1", cin, cout, 3)
    if temb:
        _lin(rng, sd, p + ".time_emb_proj", temb, cout, gain=0.3)
    _norm(rng, sd, p + ".norm2", cout)
    _conv(rng, sd, p + ".conv2", cout, cout, 3, gain=0.5)
    if cin != cout:
        _conv(rng, sd, p + ".conv_shortcut", cin, cout, 1)


def _transformer(rng, sd, p, c, ctx):
    _norm(rng, sd, p + ".norm", c)
    _conv(rng, sd, p + ".proj_in", c, c, 1)
    b = p + ".transformer_blocks.0"
    for i, kv in ((1, c), (2, ctx)):
        _norm(rng, sd, b + f".norm{i}", c)
        _lin(rng, sd, b + f".attn{i}.to_q", c, c, bias=False, gain=1.5)
        _lin(rng, sd, b + f".attn{i}.to_k", kv, c, bias=False, gain=1.5)
        _lin(rng, sd, b + f".attn{i}.to_v", kv, c, bias=False)
        _lin(rng, sd, b + f".attn{i}.to_out.0", c, c, gain=0.5)
    _norm(rng, sd, b + ".norm3", c)
    _lin(rng, sd, b + ".ff.net.0.proj", c, 8 * c)
    _lin(rng, sd, b + ".ff.net.2", 4 * c, c, gain=0.5)
    _conv(rng, sd, p + ".proj_out", c, c, 1, gain=0.5)


def _gn_gain(sd, gain):
    """Parity-stress family: the affine of every ResnetBlock2D GroupNorm (norm1 / norm2) times `gain`.  GroupNorm renormalises
    whatever reaches it, so the network does not blow up: the tensors BETWEEN a norm and the next one (norm outputs, conv
    outputs, the residual stream) grow about linearly with the gain - fp32 peak |activation| U-Net / VAE decoder: 13 / 14 at gain 1,
    92 / 122 at 16, 1.5e3 / 1.9e3 at 256, 1.2e4 / 1.5e4 at 2048, past the fp16 limit at 16 384."""
    if gain == 1.0:
        return sd
    import re
    g = np.float32(gain)
    for k in list(sd):
        if re.search(r"resnets\.\d+\.norm[12]\.(weight|bias)$", k):
            sd[k] = sd[k] * g
    return sd


def musetalk_unet_state_dict(seed: int = 4321, shapes_only: bool = False, gn_gain: float = 1.0) -> Dict[str, np.ndarray]:
    rng = _ShapeRng() if shapes_only else np.random.default_rng(seed)
    sd: Dict[str, np.ndarray] = {}
    ch = UNET_CH
    _conv(rng, sd, "conv_in", 8, ch[0], 3)
    _lin(rng, sd, "time_embedding.linear_1", ch[0], 1280)
    _lin(rng, sd, "time_embedding.linear_2", 1280, 1280)
    cin = ch[0]
    skip_ch = [ch[0]]
    for i in range(4):
        for j in range(2):
            _resnet(rng, sd, f"down_blocks.{i}.resnets.{j}", cin, ch[i], 1280)
            cin = ch[i]
            if i < 3:
                _transformer(rng, sd, f"down_blocks.{i}.attentions.{j}", ch[i], UNET_CTX)
            skip_ch.append(cin)
        if i < 3:
            _conv(rng, sd, f"down_blocks.{i}.downsamplers.0.conv", cin, cin, 3)
            skip_ch.append(cin)
    _resnet(rng, sd, "mid_block.resnets.0", cin, cin, 1280)
    _transformer(rng, sd, "mid_block.attentions.0", cin, UNET_CTX)
    _resnet(rng, sd, "mid_block.resnets.1", cin, cin, 1280)
    rev = ch[::-1]
    for i in range(4):
        for j in range(3):
            sk = skip_ch.pop()
            _resnet(rng, sd, f"up_blocks.{i}.resnets.{j}", cin + sk, rev[i], 1280)
            cin = rev[i]
            if i > 0:
                _transformer(rng, sd, f"up_blocks.{i}.attentions.{j}", cin, UNET_CTX)
        if i < 3:
            _conv(rng, sd, f"up_blocks.{i}.upsamplers.0.conv", cin, cin, 3)
    _norm(rng, sd, "conv_norm_out", cin)
    _conv(rng, sd, "conv_out", cin, 4, 3, gain=0.5)
    if not shapes_only:
        _gn_gain(sd, gn_gain)
    return _finish(sd, shapes_only)


def vae_decoder_state_dict(seed: int = 987, shapes_only: bool = False, gn_gain: float = 1.0) -> Dict[str, np.ndarray]:
    rng = _ShapeRng() if shapes_only else np.random.default_rng(seed)
    sd: Dict[str, np.ndarray] = {}
    _conv(rng, sd, "post_quant_conv", 4, 4, 1)
    top = VAE_CH[-1]
    _conv(rng, sd, "decoder.conv_in", 4, top, 3)
    _resnet(rng, sd, "decoder.mid_block.resnets.0", top, top, 0)
    a = "decoder.mid_block.attentions.0"
    _norm(rng, sd, a + ".group_norm", top)
    for nme in ("to_q", "to_k", "to_v"):
        _lin(rng, sd, a + "." + nme, top, top, gain=1.5 if nme != "to_v" else 1.0)
    _lin(rng, sd, a + ".to_out.0", top, top, gain=0.5)
    _resnet(rng, sd, "decoder.mid_block.resnets.1", top, top, 0)
    cin = top
    for i, c in enumerate(VAE_CH[::-1]):
        for j in range(3):
            _resnet(rng, sd, f"decoder.up_blocks.{i}.resnets.{j}", cin, c, 0)
            cin = c
        if i < 3:
            _conv(rng, sd, f"decoder.up_blocks.{i}.upsamplers.0.conv", cin, cin, 3)
    _norm(rng, sd, "decoder.conv_norm_out", cin)
    _conv(rng, sd, "decoder.conv_out", cin, 3, 3, gain=0.7)
    if not shapes_only:
        _gn_gain(sd, gn_gain)
    return _finish(sd, shapes_only)


def vae_encoder_state_dict(seed: int = 654, shapes_only: bool = False) -> Dict[str, np.ndarray]:
    """AutoencoderKL (sd-vae) encoder + quant_conv under diffusers' key names."""
    rng = _ShapeRng() if shapes_only else np.random.default_rng(seed)
    sd: Dict[str, np.ndarray] = {}
    _conv(rng, sd, "encoder.conv_in", 3, VAE_CH[0], 3)
    cin = VAE_CH[0]
    for i, c in enumerate(VAE_CH):
        for j in range(2):
            _resnet(rng, sd, f"encoder.down_blocks.{i}.resnets.{j}", cin, c, 0)
            cin = c
        if i < 3:
            _conv(rng, sd, f"encoder.down_blocks.{i}.downsamplers.0.conv", cin, cin, 3)
    _resnet(rng, sd, "encoder.mid_block.resnets.0", cin, cin, 0)
    a = "encoder.mid_block.attentions.0"
    _norm(rng, sd, a + ".group_norm", cin)
    for nme in ("to_q", "to_k", "to_v"):
        _lin(rng, sd, a + "." + nme, cin, cin, gain=1.5 if nme != "to_v" else 1.0)
    _lin(rng, sd, a + ".to_out.0", cin, cin, gain=0.5)
    _resnet(rng, sd, "encoder.mid_block.resnets.1", cin, cin, 0)
    _norm(rng, sd, "encoder.conv_norm_out", cin)
    _conv(rng, sd, "encoder.conv_out", cin, 8, 3, gain=0.7)
    _conv(rng, sd, "quant_conv", 8, 8, 1)
    return _finish(sd, shapes_only)


def musetalk_latents(n_frames: int = 4, seed: int = 5) -> List[np.ndarray]:
    """input_latent_list_cycle stand-in: per frame fp32 (1,8,32,32) = cat(masked, reference) VAE latents scaled by
    0.18215 (avatars/musetalk/models/vae.py:110-122)."""
    rng = np.random.default_rng(seed)
    return [(rng.standard_normal((1, 8, 32, 32)) * 0.18215 * 4).astype(np.float32) for _ in range(n_frames)]


def _soft_mask(h: int, w: int, blur: int) -> np.ndarray:
    """A lower-half box, box-blurred twice (the shape of the Gaussian-blurred masks genavatar writes,
    avatars/musetalk/utils/blending.py:129-135), stored grey as cv2.imread returns a grey PNG: (h,w,3) uint8, B = G = R."""
    m = np.zeros((h, w), np.float64)
    m[h // 2:, w // 8: w - w // 8] = 255.0
    k = np.ones(blur) / blur
    for _ in range(2):
        m = np.apply_along_axis(lambda r: np.convolve(r, k, mode="same"), 1, m)
        m = np.apply_along_axis(lambda c: np.convolve(c, k, mode="same"), 0, m)
    m8 = np.clip(np.rint(m), 0, 255).astype(np.uint8)
    return np.repeat(m8[:, :, None], 3, axis=2)


def musetalk_blend_avatar(full_hw: Tuple[int, int] = (360, 640), seed: int = 9):
    """Four frames for the MuseTalk composite (avatars/musetalk_avatar.py:154-164, myutil.py:4-25), one per case the
    composite has: 0 soft mask on a growing face box (300 px wide > 256), 1 a crop box touching the frame's left and
    bottom edges, 2 a shrinking face box (150 x 170 < 256), 3 a growing, non-square box with a hard-edged (0 / 255) mask.
    Returns (frames, masks, face_boxes (x1,y1,x2,y2), crop_boxes (x_s,y_s,x_e,y_e), preds uint8 (4,256,256,3))."""
    H, W = full_hw
    rng = np.random.default_rng(seed)
    frames = [np.ascontiguousarray(_smooth_image(rng, H, W)) for _ in range(4)]
    face_boxes = [(170, 30, 470, 330), (20, H - 230, 200, H - 30), (250, 100, 400, 270), (180, 20, 460, 340)]
    crop_boxes = [(140, 10, 500, H - 5), (0, H - 260, 230, H), (220, 60, 440, 320), (150, 5, 490, H - 2)]
    masks = []
    for i, (xs, ys, xe, ye) in enumerate(crop_boxes):
        m = _soft_mask(ye - ys, xe - xs, 15 if i != 1 else 9)
        if i == 3:
            m = np.where(m >= 128, 255, 0).astype(np.uint8)
        masks.append(m)
    preds = np.ascontiguousarray(np.stack([_smooth_image(rng, 256, 256, cells=6) for _ in range(4)]))     # C order: the device reads raw bytes
    return frames, masks, face_boxes, crop_boxes, preds


def musetalk_whisper_feats(batch: int, seed: int = 11) -> np.ndarray:
    """(B,50,384) audio feature stand-in with the value range of Whisper encoder states."""
    rng = np.random.default_rng(seed)
    return rng.standard_normal((batch, 50, 384)).astype(np.float32)


def whisper_encoder_state_dict(seed: int = 2468) -> Dict[str, np.ndarray]:
    """whisper-tiny ENCODER stand-in under transformers' key names (WhisperModel(...).encoder.state_dict()): d_model 384,
    6 heads, 4 layers, ffn 1536, 1500 positions (avatars/musetalk/whisper/audio2feature.py:15-23 loads the real one)."""
    rng = np.random.default_rng(seed)
    sd: Dict[str, np.ndarray] = {}
    D, FF = 384, 1536
    sd["conv1.weight"] = _w(rng, (D, 80, 3), 80 * 3)
    sd["conv1.bias"] = (rng.standard_normal(D) * 0.02).astype(np.float32)
    sd["conv2.weight"] = _w(rng, (D, D, 3), D * 3)
    sd["conv2.bias"] = (rng.standard_normal(D) * 0.02).astype(np.float32)
    pos = np.arange(1500)[:, None] * np.exp(-np.log(10000.0) / (D // 2 - 1) * np.arange(D // 2))[None, :]
    sd["embed_positions.weight"] = np.concatenate([np.sin(pos), np.cos(pos)], axis=1).astype(np.float32)
    for l in range(4):
        p = f"layers.{l}"
        for name in ("q_proj", "k_proj", "v_proj", "out_proj"):
            sd[f"{p}.self_attn.{name}.weight"] = _w(rng, (D, D), D, gain=0.7)
            if name != "k_proj":
                sd[f"{p}.self_attn.{name}.bias"] = (rng.standard_normal(D) * 0.02).astype(np.float32)
        for ln in ("self_attn_layer_norm", "final_layer_norm"):
            sd[f"{p}.{ln}.weight"] = (1.0 + 0.1 * rng.standard_normal(D)).astype(np.float32)
            sd[f"{p}.{ln}.bias"] = (0.05 * rng.standard_normal(D)).astype(np.float32)
        sd[f"{p}.fc1.weight"] = _w(rng, (FF, D), D)
        sd[f"{p}.fc1.bias"] = (rng.standard_normal(FF) * 0.02).astype(np.float32)
        sd[f"{p}.fc2.weight"] = _w(rng, (D, FF), FF, gain=0.7)
        sd[f"{p}.fc2.bias"] = (rng.standard_normal(D) * 0.02).astype(np.float32)
    sd["layer_norm.weight"] = (1.0 + 0.1 * rng.standard_normal(D)).astype(np.float32)
    sd["layer_norm.bias"] = (0.05 * rng.standard_normal(D)).astype(np.float32)
    return sd
