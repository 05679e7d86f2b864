"""Packed avatar bank (livetalking_amd/bank.py): round trip, alignment, copy-on-write views,
the directory packer against the layout the reference's genavatar scripts write, and the plugin fast path."""
import os
import pickle

import numpy as np
import pytest

import synth_inputs as synth
from livetalking_amd import bank


def test_wav2lip_bank_roundtrip(tmp_path):
    frames, faces, coords = synth.wav2lip_avatar(n_frames=3, full_hw=(90, 160), box=40, seed=1)
    p = str(tmp_path / "a.ltkbank")
    bank.write_wav2lip_bank(p, frames, faces, coords)
    b = bank.load_bank(p)
    f2, c2, k2 = b.as_avatar()
    assert b.kind == bank.KIND_WAV2LIP and b.n == 3 and (b.H, b.W) == (90, 160)
    assert all(np.array_equal(a, x) for a, x in zip(f2, frames)) and all(np.array_equal(a, x) for a, x in zip(c2, faces))
    assert k2 == [tuple(int(v) for v in c) for c in coords]
    assert f2.packed.shape == (3, 90, 160, 3) and c2.packed.flags["C_CONTIGUOUS"]
    # sections are page aligned (mmap -> one DMA-able host range per section)
    with open(p, "rb") as fh:
        raw = fh.read()
    assert raw[:8] == bank.MAGIC and len(raw) % bank.ALIGN == 0
    # the reference watermarks cached frames in place (base_avatar.py:417,449): views are writable, the file is not touched
    before = raw
    f2[0][0, 0, 0] ^= 0xFF
    assert f2[0].flags["WRITEABLE"]
    with open(p, "rb") as fh:
        assert fh.read() == before


def test_musetalk_bank_roundtrip(tmp_path):
    n = 3
    frames, _, _ = synth.wav2lip_avatar(n_frames=n, full_hw=(120, 200), box=40, seed=2)
    lats = synth.musetalk_latents(n)
    coords = [(60, 30, 120, 100), (61, 31, 122, 99), (59, 29, 119, 101)]
    crops = [(50, 20, 130, 110), (51, 21, 132, 109), (49, 19, 129, 111)]
    rng = np.random.default_rng(0)
    masks = [rng.integers(0, 256, (c[3] - c[1], c[2] - c[0], 3), dtype=np.uint8) for c in crops]
    p = str(tmp_path / "m.ltkbank")
    bank.write_musetalk_bank(p, frames, masks, coords, crops, lats)
    fr, mk, co, cr, la = bank.load_bank(p).as_avatar()
    assert co == coords and cr == crops
    assert all(np.array_equal(a, x) for a, x in zip(mk, masks)) and all(np.array_equal(a, x) for a, x in zip(fr, frames))
    assert all(np.array_equal(np.asarray(a), x) for a, x in zip(la, lats)) and la[0].shape == (1, 8, 32, 32)
    with pytest.raises(ValueError):
        bank.write_musetalk_bank(p, frames, masks[::-1], coords, crops, lats)     # mask / crop-box mismatch


def test_pack_avatar_dir_and_plugin_fast_path(tmp_path, monkeypatch):
    Image = pytest.importorskip("PIL.Image")
    frames, faces, coords = synth.wav2lip_avatar(n_frames=3, full_hw=(72, 128), box=32, seed=4)
    d = tmp_path / "data" / "avatars" / "av1"
    (d / "full_imgs").mkdir(parents=True)
    (d / "face_imgs").mkdir()
    for i in range(3):        # genavatar.py:124-138 writes %08d.png, BGR on disk == RGB PNG of the flipped array
        Image.fromarray(frames[i][..., ::-1]).save(d / "full_imgs" / f"{i:08d}.png")
        Image.fromarray(faces[i][..., ::-1]).save(d / "face_imgs" / f"{i:08d}.png")
    with open(d / "coords.pkl", "wb") as f:
        pickle.dump(coords, f)
    out = bank.pack_avatar_dir(str(d), kind="wav2lip")
    assert os.path.basename(out) == "bank.ltkbank"
    f2, c2, k2 = bank.load_bank(out).as_avatar()
    assert all(np.array_equal(a, x) for a, x in zip(f2, frames)) and all(np.array_equal(a, x) for a, x in zip(c2, faces))
    # the plugin's load_avatar prefers the bank (no image decoding at start-up)
    plugin = pytest.importorskip("livetalking_amd.avatars.wav2lip_avatar")
    monkeypatch.chdir(tmp_path)
    fr, fc, co = plugin.load_avatar("av1")
    assert len(fr) == 3 and fr.packed is not None and co == [tuple(int(v) for v in c) for c in coords]
