"""Resolution of the host-side base classes the plugin derives from.

Deployed as a drop-in, `avatars.base_avatar`, `avatars.
audio_features.base_asr`, `registry` and `utils.image` are the reference's own,
unmodified modules and are used as they are.  Outside the reference tree (unit
tests, bench, the GPU box) minimal stand-ins with the same names, fields and
queue protocol can be used so the plugin can be driven headless - only when
LTK_ALLOW_STANDIN=1 is set; otherwise the import fails loudly:

  AudioFrameData   avatars/base_avatar.py:57-61
  BaseAvatar       avatars/base_avatar.py:64-124 (only the fields the plugin reads)
  BaseASR          avatars/audio_features/base_asr.py:29-89
  register         registry.py:17-32
  mirror_index     utils/image.py:26-32
"""
from __future__ import annotations

import queue
from dataclasses import dataclass, field
from queue import Queue
from typing import Any, Dict

import numpy as np

import os

try:  # drop-in: the reference tree is on sys.path
    from avatars.base_avatar import AudioFrameData, BaseAvatar  # type: ignore
    from avatars.audio_features.base_asr import BaseASR  # type: ignore
    from registry import register  # type: ignore
    from utils.image import mirror_index  # type: ignore
    USING_REFERENCE_HOST = True
except ImportError as _ex:  # not inside a LiveTalking checkout (anything else - a broken checkout - propagates)
    # A deployment that reaches this branch would run sessions without TTS or stream-out: refuse unless the
    # caller asked for the headless stand-ins (tests, bench, the GPU box set LTK_ALLOW_STANDIN=1).
    if os.environ.get("LTK_ALLOW_STANDIN", "0") in ("", "0"):
        raise ImportError(
            "livetalking_amd: the reference host modules (avatars.base_avatar, avatars.audio_features.base_asr, registry, "
            "utils.image) are not importable - start from the LiveTalking checkout (scripts/run_amd.py) or set "
            "LTK_ALLOW_STANDIN=1 for the headless stand-ins (" + str(_ex) + ")") from _ex
    USING_REFERENCE_HOST = False

    @dataclass
    class AudioFrameData:
        data: Any
        type: int = 0           # 0 speech, 1 silence, >1 custom audio
        userdata: dict = field(default_factory=dict)

    _REGISTRY: Dict[str, Dict[str, type]] = {"avatar": {}}

    def register(category: str, name: str):
        def deco(cls):
            _REGISTRY.setdefault(category, {})[name] = cls
            return cls
        return deco

    def create(category: str, name: str, **kw):
        return _REGISTRY[category][name](**kw)

    def mirror_index(size: int, index: int) -> int:
        turn, res = divmod(index, size)
        return res if turn % 2 == 0 else size - res - 1

    class BaseAvatar:
        """Per-session state the plugin relies on (no TTS / stream-out here)."""

        def __init__(self, opt):
            self.opt = opt
            self.sample_rate = 16000
            self.chunk = self.sample_rate // (opt.fps * 2)
            self.sessionid = getattr(opt, "sessionid", 0)
            self.batch_size = opt.batch_size
            self.res_frame_queue = Queue(self.batch_size * 2)
            self.custom_audiotype = 0
            self.custom_index = {}
            self.speaking = False

        def put_audio_frame(self, audio_chunk, datainfo: dict = {}):
            self.asr.put_audio_frame(audio_chunk, datainfo)

        def get_avatar_length(self):
            return len(self.frame_list_cycle) if hasattr(self, "frame_list_cycle") else 1

    class BaseASR:
        """20-ms PCM chunk FIFO with silence synthesis and l/r context."""

        def __init__(self, opt, parent=None):
            self.opt = opt
            self.parent = parent
            self.fps = opt.fps
            self.sample_rate = 16000
            self.chunk = self.sample_rate // (opt.fps * 2)
            self.queue = Queue()
            self.output_queue = Queue()
            self.batch_size = opt.batch_size
            self.frames = []
            self.stride_left_size = opt.l
            self.stride_right_size = opt.r
            self.feat_queue = Queue(maxsize=2)

        def flush_talk(self):
            self.queue.queue.clear()

        def put_audio_frame(self, audio_chunk, datainfo: dict):
            self.queue.put(AudioFrameData(data=audio_chunk, type=0, userdata=datainfo))

        def get_audio_frame(self) -> AudioFrameData:
            try:
                if self.parent is not None and getattr(self.parent, "custom_audiotype", 0) > 1:
                    t = self.parent.custom_audiotype
                    return AudioFrameData(data=self.parent.get_custom_audio_stream(t), type=t, userdata={})
                return self.queue.get(block=True, timeout=0.01)
            except queue.Empty:
                return AudioFrameData(data=np.zeros(self.chunk, dtype=np.float32), type=1, userdata={})

        def get_audio_out(self) -> AudioFrameData:
            return self.output_queue.get()

        def warm_up(self):
            for _ in range(self.stride_left_size + self.stride_right_size):
                f = self.get_audio_frame()
                self.frames.append(f.data)
                self.output_queue.put(f)
            for _ in range(self.stride_left_size):
                self.output_queue.get()

        def run_step(self):
            pass

        def get_next_feat(self, block, timeout):
            return self.feat_queue.get(block, timeout)
