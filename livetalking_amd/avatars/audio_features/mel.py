"""HIP-backed drop-in for avatars/audio_features/mel.py (MelASR).

Same class name, constructor `(opt, parent)`, queue protocol and step cadence
as the reference (mel.py:34-67): every `run_step` pulls 2*batch_size 20-ms
chunks, forwards them to `output_queue`, and - once l+r chunks of context exist
- puts ONE feature batch on `feat_queue` (maxsize 2: the back-pressure) and
keeps the last l+r chunks.  The difference: the mel-spectrogram
(avatars/wav2lip/audio.py:45-51) and the (80,16) window slicing run in the
engine's HIP kernel and the feature batch is a device tensor
float32 [B][80][16] instead of a list of numpy arrays.
"""
from __future__ import annotations

import numpy as np

from ...hostshim import BaseASR

MEL_STEP_SIZE = 16      # mel.py:53
MEL_HOP = 200           # avatars/wav2lip/hparams.py:41


def window_starts(n_frames_buf: int, l: int, r: int, fps: int, n_cols: int):
    """Start column of every window for a buffer of `n_frames_buf` chunks
    (mel.py:50-63, including the tail clamp at :58-59)."""
    left = max(0, l * 80 / 50)
    mult = 80.0 / fps
    starts = []
    i = 0
    while i < (n_frames_buf - l - r) / 2:
        s = int(left + i * mult)
        if s + MEL_STEP_SIZE > n_cols:
            s = n_cols - MEL_STEP_SIZE
        starts.append(s)
        i += 1
    return starts


class MelASR(BaseASR):
    def __init__(self, opt, parent=None, engine=None):
        super().__init__(opt, parent)
        if engine is None:
            engine = getattr(parent, "engine", None) or parent.model.engine
        self.engine = engine
        import torch  # device buffers only
        self._torch = torch

    def run_step(self):
        for _ in range(self.batch_size * 2):
            audioframe = self.get_audio_frame()
            self.frames.append(audioframe.data)
            self.output_queue.put(audioframe)
        if len(self.frames) <= self.stride_left_size + self.stride_right_size:
            return
        inputs = np.concatenate(self.frames).astype(np.float32, copy=False)
        n_cols = 1 + len(inputs) // MEL_HOP          # librosa.stft(center=True) frame count
        starts = window_starts(len(self.frames), self.stride_left_size, self.stride_right_size, self.fps, n_cols)
        torch = self._torch
        feat = torch.empty((len(starts), 80, MEL_STEP_SIZE), dtype=torch.float32,
                           device=self.engine.torch_device)
        self.engine.mel_step(inputs, starts, feat.data_ptr())
        self.feat_queue.put(feat)
        self.frames = self.frames[-(self.stride_left_size + self.stride_right_size):]
