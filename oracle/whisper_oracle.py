"""Oracle: MuseTalk audio features (Whisper-tiny encoder states + chunk slicing), CPU.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference path:
  avatars/audio_features/whisper.py:58-76   WhisperASR.run_step (52 chunks -> audio2feat -> _feature2chunks)
  avatars/audio_features/whisper.py:35-56   _feature2chunks
  avatars/audio_features/base_asr.py:91-133 _get_sliced_feature (index clamping)
  avatars/musetalk/whisper/audio2feature.py:15-23,106-117  Audio2Feature: AutoFeatureExtractor + WhisperModel.encoder
      (output_hidden_states=True) -> torch.stack(hidden_states, dim=2) -> (1500, 5, 384)

Third-party arithmetic: `transformers` (WhisperFeatureExtractor, WhisperModel) IS installed in this image, so this
oracle CALLS it instead of restating it - the checker is the library the reference itself calls.  The model is a
whisper-tiny-shaped WhisperModel with seeded random weights (no checkpoint exists in the reference tree or here).
"""
from __future__ import annotations

import numpy as np
import torch


def tiny_whisper(seed: int = 0):
    from transformers import WhisperConfig, WhisperModel
    cfg = WhisperConfig(d_model=384, encoder_layers=4, encoder_attention_heads=6, encoder_ffn_dim=1536, decoder_layers=4,
                        decoder_attention_heads=6, decoder_ffn_dim=1536, num_mel_bins=80, max_source_positions=1500,
                        vocab_size=51865)
    torch.manual_seed(seed)
    return WhisperModel(cfg).eval()


def input_features(wav: np.ndarray) -> torch.Tensor:
    """audio2feature.py:107-111 -> (1, 80, 3000) float32."""
    from transformers import WhisperFeatureExtractor
    fe = WhisperFeatureExtractor()
    return fe(np.asarray(wav, dtype=np.float32), return_tensors="pt", sampling_rate=16000).input_features


def audio2feat(model, wav: np.ndarray):
    """audio2feature.py:106-117 on CPU/fp32 -> (1500, 5, 384) and the list of hidden states."""
    feats = input_features(wav)
    with torch.no_grad():
        hs = model.encoder(feats, output_hidden_states=True).hidden_states
    return torch.stack(hs, dim=2).squeeze(0).numpy(), [h.squeeze(0) for h in hs], feats


def get_sliced_feature(feature_array, vid_idx, audio_feat_win, feature_idx_multiplier=1.0):
    """base_asr.py:91-133."""
    length = feature_array.shape[0]
    center_idx = int(vid_idx * feature_idx_multiplier)
    left = int(center_idx - audio_feat_win[0] * feature_idx_multiplier)
    right = int(center_idx + audio_feat_win[1] * feature_idx_multiplier)
    sel = []
    for idx in range(left, right):
        idx = min(length - 1, max(0, idx))
        sel.append(feature_array[idx])
    return np.asarray(sel)


def feature2chunks(feature_array, batch_size, l: int = 10):
    """whisper.py:35-56 as run_step calls it (whisper.py:71-73): win [0,5], start l/2, multiplier 2."""
    chunks = []
    for i in range(batch_size):
        sel = get_sliced_feature(feature_array, i + l / 2, [0, 5], 2)
        chunks.append(sel.reshape(-1, 384))
    return chunks
