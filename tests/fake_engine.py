"""CPU stand-in for livetalking_amd.engine.Engine (TEST INFRASTRUCTURE): same method surface the plugin modules call,
arithmetic by the oracle.  It lets the CPU suite run the plugin behind the reference's unmodified render loop
(oracle/ref_loop.py, mode plugin-fake) without a GPU; it is never importable from the product package.

Pointers are host addresses here (`torch_device` is the CPU): the plugin allocates its "device" tensors on the CPU and
hands their data_ptr()s over exactly as it does with HBM addresses."""
from __future__ import annotations

import ctypes
import threading

import numpy as np
import torch

from oracle import mel_oracle, paste_oracle, plugin_oracle


def _view(ptr: int, shape, dtype):
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    buf = (ctypes.c_char * n).from_address(int(ptr))
    return np.frombuffer(buf, dtype=dtype).reshape(shape)


class FakeEngine:
    def __init__(self, net="tiny", sd_np=None, device=0, max_frames=256):
        self.device = device
        self.max_frames = max_frames
        self.net = net
        if net == "tiny":
            from oracle.ref_loop import tiny_lip
            self._model = tiny_lip()
        else:
            from oracle import wav2lip_oracle
            sd = {k: torch.from_numpy(v) for k, v in sd_np.items()}
            self._model = lambda mel, img: wav2lip_oracle.forward(sd, mel, img, None)
        self._avatars = {}
        self._next = 1
        self._lock = threading.Lock()
        self.calls = {"mel_step": 0, "wav2lip_infer": 0, "paste_back": 0, "frames": 0}
        self.closed = False

    @property
    def torch_device(self):
        return torch.device("cpu")

    def close(self):
        self.closed = True

    def register_avatar(self, face_list, frame_list, coord_list):
        with self._lock:
            aid = self._next
            self._next += 1
            self._avatars[aid] = ([np.asarray(f) for f in face_list], [np.asarray(f) for f in frame_list],
                                  [tuple(int(v) for v in c) for c in coord_list])
            return aid

    def release_avatar(self, aid):
        with self._lock:
            self._avatars.pop(aid, None)

    def mel_step(self, pcm, win_start, d_out_ptr, stream=0):
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        mel = mel_oracle.melspectrogram(pcm)                 # (80, n_cols) float64, the reference's own arithmetic
        out = _view(d_out_ptr, (len(win_start), 80, 16), np.float32)
        for i, s in enumerate(win_start):
            out[i] = mel[:, s:s + 16]
        self.calls["mel_step"] += 1

    def wav2lip_infer(self, reqs, stream=0):
        for aid, index, batch, mel_ptr, pred_ptr in reqs:
            faces, _, _ = self._avatars[aid]
            feats = list(_view(mel_ptr, (batch, 80, 16), np.float32).astype(np.float64))
            mel_t, img_t = plugin_oracle.pack_inputs(faces, index, batch, feats)
            with torch.no_grad():
                pred = self._model(mel_t, img_t)
            frames = (pred.cpu().numpy().transpose(0, 2, 3, 1) * 255.).astype(np.uint8)   # wav2lip_avatar.py:138,145 truncation
            _view(pred_ptr, (batch, 256, 256, 3), np.uint8)[...] = frames
            self.calls["frames"] += batch
        self.calls["wav2lip_infer"] += 1

    def paste_back(self, aid, idx, d_pred_ptr, out, stream=0):
        _, frames, coords = self._avatars[aid]
        pred = _view(d_pred_ptr, (256, 256, 3), np.uint8)
        out[...] = paste_oracle.paste_back_frame(pred.astype(np.float32), frames[idx], coords[idx])
        self.calls["paste_back"] += 1

    def wav2lip_forward_host(self, mel, face6):
        with torch.no_grad():
            return self._model(torch.from_numpy(np.asarray(mel, np.float32)).reshape(-1, 1, 80, 16),
                               torch.from_numpy(np.asarray(face6, np.float32))).numpy()
