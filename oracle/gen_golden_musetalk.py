#!/usr/bin/env python3
"""Pin the MuseTalk HOST-side restatements against the reference's own code (run in the build container).

TEST INFRASTRUCTURE ONLY.   python -m oracle.gen_golden_musetalk [--ref /root/reference]

What runs from the reference, unmodified (imported, never copied):
  avatars.musetalk.models.unet.PositionalEncoding            (unet.py:12-27; `diffusers` stubbed: only the class import)
  avatars.audio_features.whisper.WhisperASR._feature2chunks  (whisper.py:35-56) + BaseASR._get_sliced_feature
                                                             (base_asr.py:91-133), via WhisperASR.run_step's arguments
  avatars.musetalk_avatar: mirror_index use                  (utils/image.py:26-32)
The U-Net / VAE arithmetic itself lives in `diffusers` (absent): see oracle/musetalk_oracle.py (PARITY UNPINNED).
Writes tests/golden/musetalk_host_golden.npz after asserting that the oracle restatements agree.
"""
from __future__ import annotations

import argparse
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)

from oracle import gen_golden, musetalk_oracle, whisper_oracle  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    args = ap.parse_args()
    gen_golden.install_stubs()
    gen_golden._stub("diffusers", UNet2DConditionModel=object, AutoencoderKL=object)
    gen_golden._stub("torchvision")
    gen_golden._stub("torchvision.transforms", Normalize=lambda **k: None)
    sys.path.insert(0, args.ref)
    os.chdir(tempfile.mkdtemp(prefix="ltk_golden_mt_"))

    # ---- PositionalEncoding (unet.py:12-27)
    from avatars.musetalk.models.unet import PositionalEncoding
    pe = PositionalEncoding(d_model=384)
    x = torch.from_numpy(np.random.default_rng(0).standard_normal((2, 50, 384)).astype(np.float32))
    ref_pe = pe(x).numpy()
    mine = musetalk_oracle.positional_encoding(x).numpy()
    assert np.abs(ref_pe - mine).max() < 1e-6, "PositionalEncoding restatement drifted"
    pe_table = pe(torch.zeros(1, 50, 384)).numpy()[0]

    # ---- whisper chunk slicing (whisper.py:35-56,71-73; base_asr.py:91-133)
    from avatars.audio_features.whisper import WhisperASR
    asr = WhisperASR.__new__(WhisperASR)          # no queues / opt needed for the slicing helper
    feat = np.arange(1500 * 5 * 384, dtype=np.float32).reshape(1500, 5, 384)
    chunks = asr._feature2chunks(feature_array=feat, batch_size=16, audio_feat_win=[0, 5], start=10 / 2, feature_idx_multiplier=2)
    ref_chunks = np.stack(chunks)
    mine_chunks = np.stack(whisper_oracle.feature2chunks(feat, 16, 10))
    assert ref_chunks.shape == (16, 50, 384) and np.array_equal(ref_chunks, mine_chunks), "chunk slicing restatement drifted"
    rows = (ref_chunks[:, :, 0] // (5 * 384)).astype(np.int32)[:, ::5]      # encoder row of every 5-state group

    # ---- edge: clamping at the end of the feature array (short arrays)
    short = feat[:40]
    ref_short = np.stack(asr._feature2chunks(feature_array=short, batch_size=16, audio_feat_win=[0, 5], start=5.0, feature_idx_multiplier=2))
    assert np.array_equal(ref_short, np.stack(whisper_oracle.feature2chunks(short, 16, 10)))
    rows_short = (ref_short[:, :, 0] // (5 * 384)).astype(np.int32)[:, ::5]

    np.savez_compressed(os.path.join(args.out, "musetalk_host_golden.npz"), pe_table=pe_table.astype(np.float32), chunk_rows=rows,
                        chunk_rows_short=rows_short)
    print("wrote musetalk_host_golden.npz: pe_table", pe_table.shape, "chunk_rows", rows.shape, rows[0], rows[-1])


if __name__ == "__main__":
    main()
