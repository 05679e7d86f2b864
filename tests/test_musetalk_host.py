"""CPU checks of the MuseTalk oracle pieces that ARE pinned to the reference's own code
(oracle/gen_golden_musetalk.py imported avatars.musetalk.models.unet.PositionalEncoding and
avatars.audio_features.whisper.WhisperASR._feature2chunks and wrote tests/golden/musetalk_host_golden.npz), plus
properties of the restated blendLinear paste-back and of the VAE-decoder oracle."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from livetalking_amd import synth  # noqa: E402
from oracle import musetalk_oracle as M  # noqa: E402
from oracle import paste_oracle, whisper_oracle  # noqa: E402


def test_positional_encoding_vs_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "musetalk_host_golden.npz"))
    pe = M.positional_encoding(torch.zeros(1, 50, 384)).numpy()[0]
    assert np.abs(pe - g["pe_table"]).max() < 1e-6


def test_chunk_rows_vs_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "musetalk_host_golden.npz"))
    feat = np.arange(1500 * 5 * 384, dtype=np.float32).reshape(1500, 5, 384)
    got = np.stack(whisper_oracle.feature2chunks(feat, 16, 10))
    rows = (got[:, :, 0] // (5 * 384)).astype(np.int32)[:, ::5]
    assert np.array_equal(rows, g["chunk_rows"])
    # frame i takes encoder rows 2(i+5) .. 2(i+5)+9, five hidden states per row
    assert rows[0, 0] == 10 and rows[15, 9] == 49
    short = np.stack(whisper_oracle.feature2chunks(feat[:40], 16, 10))
    assert np.array_equal((short[:, :, 0] // (5 * 384)).astype(np.int32)[:, ::5], g["chunk_rows_short"])


def test_blend_paste_properties():
    rng = np.random.default_rng(0)
    pred = rng.integers(0, 256, (256, 256, 3), dtype=np.uint8)
    frame = rng.integers(0, 256, (180, 320, 3), dtype=np.uint8)
    bbox, crop = (120, 40, 220, 150), (100, 20, 240, 170)
    zero = np.zeros((150, 140, 3), np.uint8)
    out0 = paste_oracle.paste_blend_frame(pred, frame, bbox, zero, crop)
    assert np.array_equal(out0, frame)                       # mask 0: the cached frame survives untouched
    full = np.full((150, 140, 3), 255, np.uint8)
    out1 = paste_oracle.paste_blend_frame(pred, frame, bbox, full, crop)
    face = paste_oracle.resize_linear_u8(pred, (100, 110))
    assert np.array_equal(out1[40:150, 120:220], face)       # mask 255 inside the face box: the resized prediction
    outside = np.ones(frame.shape[:2], bool)
    outside[20:170, 100:240] = False
    assert np.array_equal(out1[outside], frame[outside])     # nothing outside the crop box ever changes
    assert out1.flags["C_CONTIGUOUS"] and out1.dtype == np.uint8


def test_vae_decoder_oracle_runs_and_is_deterministic():
    vsd = {k: torch.from_numpy(v) for k, v in synth.vae_decoder_state_dict().items()}
    z = torch.from_numpy(synth.musetalk_latents(1)[0][:, :4])
    with torch.no_grad():
        a = M.decode_latents(vsd, z)
        b = M.decode_latents(vsd, z)
    assert a.shape == (1, 256, 256, 3) and a.dtype == np.uint8 and np.array_equal(a, b)
    assert a.std() > 5.0                                     # seeded weights give a non-degenerate image
