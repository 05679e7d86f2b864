"""Oracle: the Wav2Lip plugin's per-batch host logic, numpy + torch fp32 on CPU.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates
  avatars/wav2lip_avatar.py:116-139  LipReal.inference_batch
      (bank gather by mirror_index, lower-half mask, 6-channel concat, /255.,
       NCHW transpose, forward, *255, NHWC)
  avatars/wav2lip_avatar.py:90-96    warm_up input shapes
Pinned by gen_golden.py against the reference's own LipReal.inference_batch
(imported from the upstream checkout with its third-party imports stubbed).
"""
from __future__ import annotations

from typing import Dict, Sequence

import numpy as np
import torch

from . import wav2lip_oracle as W
from .paste_oracle import mirror_index


def pack_inputs(face_list_cycle: Sequence[np.ndarray], index: int, batch_size: int,
                audiofeat_batch) -> tuple:
    """wav2lip_avatar.py:119-134 -> (mel (B,1,80,16) fp32, img (B,6,256,256) fp32)."""
    length = len(face_list_cycle)
    img_batch = []
    for i in range(batch_size):
        idx = mirror_index(length, index + i)
        img_batch.append(face_list_cycle[idx])
    img_batch = np.asarray(img_batch)
    audiofeat = np.asarray(audiofeat_batch)
    img_masked = img_batch.copy()
    img_masked[:, img_batch.shape[1] // 2:] = 0
    img6 = np.concatenate((img_masked, img_batch), axis=3) / 255.
    audiofeat = np.reshape(audiofeat, [len(audiofeat), audiofeat.shape[1], audiofeat.shape[2], 1])
    img_t = torch.FloatTensor(np.transpose(img6, (0, 3, 1, 2)))
    mel_t = torch.FloatTensor(np.transpose(audiofeat, (0, 3, 1, 2)))
    return mel_t, img_t


def inference_batch(sd: Dict[str, torch.Tensor], face_list_cycle, index: int, batch_size: int,
                    audiofeat_batch) -> np.ndarray:
    """wav2lip_avatar.py:116-139 -> float32 (B,256,256,3) BGR in [0,255]."""
    mel_t, img_t = pack_inputs(face_list_cycle, index, batch_size, audiofeat_batch)
    pred = W.forward(sd, mel_t, img_t)
    return pred.cpu().numpy().transpose(0, 2, 3, 1) * 255.
