"""HIP-backed drop-in for avatars/audio_features/whisper.py (WhisperASR).

Same class name, constructor `(opt, parent, audio_processor)`, queue protocol and step cadence as the reference
(whisper.py:58-76): every `run_step` pulls 2*batch_size 20-ms chunks, forwards them to `output_queue`, and - once
l+r chunks of context exist - puts ONE feature batch on `feat_queue` and keeps the last l+r chunks.  The
difference: audio2feat (log-mel + Whisper encoder with its 5 hidden states, audio2feature.py:106-117) and the
(50,384) chunk slicing (whisper.py:35-56) run on the engine and the feature batch is ONE device tensor
float32 [B][50][384] instead of a list of numpy arrays.
"""
from __future__ import annotations

import numpy as np

from ...hostshim import BaseASR


# what `whisper_logmel_kernel` computes = WhisperFeatureExtractor's defaults = whisper-tiny's preprocessor_config.json
FEATURE_EXTRACTOR = {"feature_size": 80, "sampling_rate": 16000, "hop_length": 160, "chunk_length": 30, "n_fft": 400}
# the encoder program's shape (whisper-tiny; config.json beside the checkpoint)
ENCODER_SHAPE = {"d_model": 384, "encoder_layers": 4, "encoder_attention_heads": 6, "encoder_ffn_dim": 1536, "num_mel_bins": 80,
                 "max_source_positions": 1500}


def check_whisper_dir(model_path: str) -> None:
    """The reference builds its front end from the directory's files (AutoFeatureExtractor / WhisperModel.from_pretrained,
    audio2feature.py:20-21); the engine's log-mel kernel and encoder program are fixed, so a directory that asks for
    anything else is an error."""
    import json
    import os
    for fname, want in (("preprocessor_config.json", FEATURE_EXTRACTOR), ("config.json", ENCODER_SHAPE)):
        p = os.path.join(model_path, fname)
        if not os.path.exists(p):
            continue
        with open(p) as f:
            cfg = json.load(f)
        bad = [f"{k}: file has {cfg[k]!r}, engine is built for {v!r}" for k, v in want.items() if k in cfg and cfg[k] != v]
        if bad:
            raise ValueError(f"{p}: " + "; ".join(bad))


class Audio2Feature:
    """avatars/musetalk/whisper/audio2feature.py:15-23: owns the Whisper encoder; here it lives in the engine."""

    def __init__(self, engine, encoder_state_dict=None, model_path="./models/whisper"):
        self.engine = engine
        if encoder_state_dict is None:
            from transformers import WhisperModel   # the checkpoint reader the reference uses
            check_whisper_dir(model_path)
            encoder_state_dict = WhisperModel.from_pretrained(model_path).encoder.state_dict()
        engine.load_whisper(encoder_state_dict)


class WhisperASR(BaseASR):
    def __init__(self, opt, parent, audio_processor: Audio2Feature):
        super().__init__(opt, parent)
        self.audio_processor = audio_processor
        self.engine = audio_processor.engine
        import torch  # device buffers only
        self._torch = torch

    def run_step(self):
        for _ in range(self.batch_size * 2):
            audio_frame = self.get_audio_frame()
            self.frames.append(audio_frame.data)
            self.output_queue.put(audio_frame)
        if len(self.frames) <= self.stride_left_size + self.stride_right_size:
            return
        inputs = np.concatenate(self.frames).astype(np.float32, copy=False)
        torch = self._torch
        feat = torch.empty((self.batch_size, 50, 384), dtype=torch.float32, device=self.engine.torch_device)
        # whisper.py:71-73: audio_feat_win [0,5], start l/2, multiplier 2 -> rows [2*(i + l/2), +10)
        first_row = int((self.stride_left_size / 2) * 2)
        self.engine.whisper_step(inputs, self.batch_size, first_row, feat.data_ptr(), row_step=2, rows=10)
        self.feat_queue.put(feat)
        self.frames = self.frames[-(self.stride_left_size + self.stride_right_size):]
