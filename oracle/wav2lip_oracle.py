"""Oracle: Wav2Lip-256 generator forward, plain torch fp32 on CPU.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates, as a flat layer table + functional ops, the module graph of
  avatars/wav2lip/models/wav2lip_v2.py:8-91   (layer definitions)
  avatars/wav2lip/models/wav2lip_v2.py:123-163 (forward: audio encoder, face
      encoder with feature stack, decoder with torch.cat skips, output block)
  avatars/wav2lip/models/conv.py:5-19  (Conv2d  = conv -> BN(eval) -> [+x] -> ReLU)
  avatars/wav2lip/models/conv.py:33-44 (Conv2dTranspose = convT -> BN(eval) -> ReLU)

Pinning: `oracle/gen_golden.py` runs the reference's own `Wav2Lip` nn.Module
(imported from the upstream checkout) on the same seeded state-dict and inputs
and asserts agreement before writing tests/golden/wav2lip_*.npz; the committed
fixtures carry the reference module's outputs, not this file's.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F


@dataclass(frozen=True)
class LayerSpec:
    prefix: str          # state_dict prefix, e.g. "face_encoder_blocks.1.0"
    kind: str            # "conv" | "convT"
    cin: int
    cout: int
    k: Tuple[int, int]
    stride: Tuple[int, int]
    pad: Tuple[int, int]
    out_pad: Tuple[int, int] = (0, 0)
    residual: bool = False


def _c(prefix, cin, cout, k, s, p, residual=False):
    k = (k, k) if isinstance(k, int) else k
    s = (s, s) if isinstance(s, int) else s
    p = (p, p) if isinstance(p, int) else p
    return LayerSpec(prefix, "conv", cin, cout, k, s, p, (0, 0), residual)


def _t(prefix, cin, cout, k, s, p, op=0):
    return LayerSpec(prefix, "convT", cin, cout, (k, k), (s, s), (p, p), (op, op), False)


# wav2lip_v2.py:41-58
AUDIO_ENCODER: List[LayerSpec] = [
    _c("audio_encoder.0", 1, 32, 3, 1, 1),
    _c("audio_encoder.1", 32, 32, 3, 1, 1, True),
    _c("audio_encoder.2", 32, 32, 3, 1, 1, True),
    _c("audio_encoder.3", 32, 64, 3, (3, 1), 1),
    _c("audio_encoder.4", 64, 64, 3, 1, 1, True),
    _c("audio_encoder.5", 64, 64, 3, 1, 1, True),
    _c("audio_encoder.6", 64, 128, 3, 3, 1),
    _c("audio_encoder.7", 128, 128, 3, 1, 1, True),
    _c("audio_encoder.8", 128, 128, 3, 1, 1, True),
    _c("audio_encoder.9", 128, 256, 3, (3, 2), 1),
    _c("audio_encoder.10", 256, 256, 3, 1, 1, True),
    _c("audio_encoder.11", 256, 512, 3, 1, 0),
    _c("audio_encoder.12", 512, 512, 1, 1, 0),
]

# wav2lip_v2.py:12-39
FACE_ENCODER_BLOCKS: List[List[LayerSpec]] = [
    [_c("face_encoder_blocks.0.0", 6, 16, 7, 1, 3)],
    [_c("face_encoder_blocks.1.0", 16, 32, 3, 2, 1),
     _c("face_encoder_blocks.1.1", 32, 32, 3, 1, 1, True),
     _c("face_encoder_blocks.1.2", 32, 32, 3, 1, 1, True)],
    [_c("face_encoder_blocks.2.0", 32, 64, 3, 2, 1),
     _c("face_encoder_blocks.2.1", 64, 64, 3, 1, 1, True),
     _c("face_encoder_blocks.2.2", 64, 64, 3, 1, 1, True),
     _c("face_encoder_blocks.2.3", 64, 64, 3, 1, 1, True)],
    [_c("face_encoder_blocks.3.0", 64, 128, 3, 2, 1),
     _c("face_encoder_blocks.3.1", 128, 128, 3, 1, 1, True),
     _c("face_encoder_blocks.3.2", 128, 128, 3, 1, 1, True)],
    [_c("face_encoder_blocks.4.0", 128, 256, 3, 2, 1),
     _c("face_encoder_blocks.4.1", 256, 256, 3, 1, 1, True),
     _c("face_encoder_blocks.4.2", 256, 256, 3, 1, 1, True)],
    [_c("face_encoder_blocks.5.0", 256, 512, 3, 2, 1),
     _c("face_encoder_blocks.5.1", 512, 512, 3, 1, 1, True)],
    [_c("face_encoder_blocks.6.0", 512, 512, 3, 2, 1),
     _c("face_encoder_blocks.6.1", 512, 512, 3, 1, 1, True)],
    [_c("face_encoder_blocks.7.0", 512, 512, 4, 1, 0),
     _c("face_encoder_blocks.7.1", 512, 512, 1, 1, 0)],
]

# wav2lip_v2.py:60-87
FACE_DECODER_BLOCKS: List[List[LayerSpec]] = [
    [_c("face_decoder_blocks.0.0", 512, 512, 1, 1, 0)],
    [_t("face_decoder_blocks.1.0", 1024, 512, 4, 1, 0),
     _c("face_decoder_blocks.1.1", 512, 512, 3, 1, 1, True)],
    [_t("face_decoder_blocks.2.0", 1024, 512, 3, 2, 1, 1),
     _c("face_decoder_blocks.2.1", 512, 512, 3, 1, 1, True)],
    [_t("face_decoder_blocks.3.0", 1024, 512, 3, 2, 1, 1),
     _c("face_decoder_blocks.3.1", 512, 512, 3, 1, 1, True),
     _c("face_decoder_blocks.3.2", 512, 512, 3, 1, 1, True)],
    [_t("face_decoder_blocks.4.0", 768, 384, 3, 2, 1, 1),
     _c("face_decoder_blocks.4.1", 384, 384, 3, 1, 1, True),
     _c("face_decoder_blocks.4.2", 384, 384, 3, 1, 1, True)],
    [_t("face_decoder_blocks.5.0", 512, 256, 3, 2, 1, 1),
     _c("face_decoder_blocks.5.1", 256, 256, 3, 1, 1, True),
     _c("face_decoder_blocks.5.2", 256, 256, 3, 1, 1, True)],
    [_t("face_decoder_blocks.6.0", 320, 128, 3, 2, 1, 1),
     _c("face_decoder_blocks.6.1", 128, 128, 3, 1, 1, True),
     _c("face_decoder_blocks.6.2", 128, 128, 3, 1, 1, True)],
    [_t("face_decoder_blocks.7.0", 160, 64, 3, 2, 1, 1),
     _c("face_decoder_blocks.7.1", 64, 64, 3, 1, 1, True),
     _c("face_decoder_blocks.7.2", 64, 64, 3, 1, 1, True)],
]

# wav2lip_v2.py:89-91: Conv2d(80,32,3,1,1) ; nn.Conv2d(32,3,1,1,0) ; Sigmoid
OUTPUT_CONV = _c("output_block.0", 80, 32, 3, 1, 1)
OUTPUT_HEAD_PREFIX = "output_block.1"  # plain nn.Conv2d(32, 3, 1): keys .weight/.bias

BN_EPS = 1e-5  # torch.nn.BatchNorm2d default, conv.py:9,38


def all_block_layers() -> List[LayerSpec]:
    out = list(AUDIO_ENCODER)
    for blk in FACE_ENCODER_BLOCKS:
        out += blk
    for blk in FACE_DECODER_BLOCKS:
        out += blk
    out.append(OUTPUT_CONV)
    return out


def macs_per_frame() -> int:
    """Conv/convT multiply-accumulates per 256x256 output frame (SURVEY App. A:
    27.789 GMAC).  Recomputed here from the table so tests can pin it."""
    total = 0

    def run(layer, hw):
        h, w = hw
        kh, kw = layer.k
        if layer.kind == "conv":
            ho = (h + 2 * layer.pad[0] - kh) // layer.stride[0] + 1
            wo = (w + 2 * layer.pad[1] - kw) // layer.stride[1] + 1
            m = layer.cin * layer.cout * kh * kw * ho * wo
        else:
            ho = (h - 1) * layer.stride[0] - 2 * layer.pad[0] + kh + layer.out_pad[0]
            wo = (w - 1) * layer.stride[1] - 2 * layer.pad[1] + kw + layer.out_pad[1]
            m = layer.cin * layer.cout * kh * kw * h * w
        return m, (ho, wo)

    hw = (80, 16)
    for l in AUDIO_ENCODER:
        m, hw = run(l, hw)
        total += m
    hw = (256, 256)
    for blk in FACE_ENCODER_BLOCKS:
        for l in blk:
            m, hw = run(l, hw)
            total += m
    hw = (1, 1)
    for blk in FACE_DECODER_BLOCKS:
        for l in blk:
            m, hw = run(l, hw)
            total += m
    m, hw = run(OUTPUT_CONV, hw)
    total += m
    total += 32 * 3 * hw[0] * hw[1]
    return total


def _block(x: torch.Tensor, sd: Dict[str, torch.Tensor], l: LayerSpec) -> torch.Tensor:
    """conv.py:15-19 / conv.py:41-44 in eval mode."""
    w = sd[l.prefix + ".conv_block.0.weight"]
    b = sd[l.prefix + ".conv_block.0.bias"]
    if l.kind == "conv":
        y = F.conv2d(x, w, b, stride=l.stride, padding=l.pad)
    else:
        y = F.conv_transpose2d(x, w, b, stride=l.stride, padding=l.pad, output_padding=l.out_pad)
    y = F.batch_norm(
        y,
        sd[l.prefix + ".conv_block.1.running_mean"],
        sd[l.prefix + ".conv_block.1.running_var"],
        sd[l.prefix + ".conv_block.1.weight"],
        sd[l.prefix + ".conv_block.1.bias"],
        training=False, eps=BN_EPS)
    if l.residual:
        y = y + x
    return F.relu(y)


@torch.no_grad()
def forward(sd: Dict[str, torch.Tensor], mel: torch.Tensor, face: torch.Tensor,
            taps: Dict[str, torch.Tensor] | None = None) -> torch.Tensor:
    """wav2lip_v2.py:123-163 for 4-D inputs (the only case the render loop uses).

    mel  (B,1,80,16) fp32, face (B,6,256,256) fp32 in [0,1] -> (B,3,256,256) in (0,1).
    If `taps` is a dict, every layer's output is stored under its prefix
    (used for per-layer parity of the HIP conv kernels).
    """
    x = mel
    for l in AUDIO_ENCODER:
        x = _block(x, sd, l)
        if taps is not None:
            taps[l.prefix] = x
    audio_embedding = x  # (B,512,1,1)

    feats = []
    x = face
    for blk in FACE_ENCODER_BLOCKS:
        for l in blk:
            x = _block(x, sd, l)
            if taps is not None:
                taps[l.prefix] = x
        feats.append(x)

    x = audio_embedding
    for blk in FACE_DECODER_BLOCKS:
        for l in blk:
            x = _block(x, sd, l)
            if taps is not None:
                taps[l.prefix] = x
        x = torch.cat((x, feats[-1]), dim=1)  # wav2lip_v2.py:146
        feats.pop()

    x = _block(x, sd, OUTPUT_CONV)
    if taps is not None:
        taps[OUTPUT_CONV.prefix] = x
    x = F.conv2d(x, sd[OUTPUT_HEAD_PREFIX + ".weight"], sd[OUTPUT_HEAD_PREFIX + ".bias"])
    return torch.sigmoid(x)
