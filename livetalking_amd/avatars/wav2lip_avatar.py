"""HIP-backed drop-in for avatars/wav2lip_avatar.py.

Module contract the reference's app.py relies on (app.py:128-151,99):
  load_model(path) -> model handle        (wav2lip_avatar.py:59-70)
  load_avatar(avatar_id) -> avatar tuple  (wav2lip_avatar.py:72-88)
  warm_up(batch_size, model, modelres)    (wav2lip_avatar.py:90-96)
  @register("avatar", "wav2lip") class LipReal(BaseAvatar) with
  inference_batch(index, audiofeat_batch) and paste_back_frame(pred_frame, idx)
                                          (wav2lip_avatar.py:98-147)

What changes underneath: the model handle owns one `Engine` (libltk_hip.so) per
GPU and pins every session to the least-loaded one; the avatar bank is uploaded to HBM once per avatar;
`inference_batch` returns device handles (uint8 256x256x3 crops, already
truncated the way paste_back_frame's astype(uint8) does) instead of float
numpy frames; `paste_back_frame` composites on the GPU and returns the same
writable C-contiguous uint8 (H,W,3) BGR array the reference returns.
"""
from __future__ import annotations

import glob
import os
import pickle
import threading
import weakref

import numpy as np

from ..egress import SRC_WAV2LIP, DeviceEgressMixin, FrameGroup
from ..engine import Engine
from ..hostshim import BaseAvatar, mirror_index, register  # noqa: F401  (mirror_index: part of the reference module's namespace)
from ..scheduler import get_scheduler
from ..sharding import EnginePool, visible_devices
from .audio_features.mel import MelASR

_PASTE_BATCH = os.environ.get("LTK_PASTE_BATCH", "1") != "0"     # 0: one composite + one pageable copy per paste_back_frame call


class Wav2LipModel:
    """Opaque `model` object handed back to app.py; process-global, shared by sessions (app.py:62-63).  Owns one engine
    per GPU (sharding.EnginePool); a session is pinned to one of them when it is constructed."""

    def __init__(self, engine_or_pool):
        if isinstance(engine_or_pool, EnginePool):
            self.pool = engine_or_pool
        else:                                   # a single engine (tests hand in a fake one)
            self.pool = EnginePool([getattr(engine_or_pool, "device", 0)], lambda d: engine_or_pool)
        self.engines = self.pool.engines
        self._avatars = {}          # (id(face_list), engine slot) -> engine avatar id
        self._lock = threading.Lock()

    @property
    def engine(self) -> Engine:
        """The first engine (single-GPU callers, warm-up of a one-GPU deployment)."""
        return self.engines[0]

    def place(self, session) -> int:
        slot = self.pool.place(id(session))
        weakref.finalize(session, self.pool.release, id(session))      # session_manager.remove_session just drops the object
        return slot

    def avatar_id(self, avatar, slot: int = 0) -> int:
        frame_list, face_list, coord_list = avatar
        key = (id(face_list), slot)
        with self._lock:
            aid = self._avatars.get(key)
            if aid is None:                     # first session of this avatar on this GPU: upload the bank replica
                aid = self.engines[slot].register_avatar(face_list, frame_list, coord_list)
                self._avatars[key] = aid
            return aid

    def eval(self):
        return self


def _state_dict_from_checkpoint(path):
    import torch
    checkpoint = torch.load(path, map_location="cpu")
    s = checkpoint["state_dict"]
    return {k.replace("module.", ""): v for k, v in s.items()}


def load_model(path, state_dict=None, max_frames=None, device=None):
    """`path` is the reference's ./models/wav2lip.pth; `state_dict` may be passed
    directly (tests / bench use seeded synthetic weights: there is no checkpoint
    in the reference tree).  One engine per GPU of LTK_DEVICES (default: every visible GPU), or just `device`."""
    if state_dict is None:
        state_dict = _state_dict_from_checkpoint(path)
    if max_frames is None:
        max_frames = int(os.environ.get("LTK_MAX_FRAMES", "256"))

    def factory(dev):
        eng = Engine(dev)
        eng.load_wav2lip(state_dict, max_frames=max_frames)
        return eng

    devices = visible_devices() if device is None else [int(device)]
    cap = int(os.environ.get("LTK_SESSIONS_PER_GPU", "0")) or (1 << 30)
    return Wav2LipModel(EnginePool(devices, factory, capacity_per_gpu=cap))


def read_imgs(img_list):
    import cv2  # same third-party reader the reference uses (utils/image.py:14-24)
    return [cv2.imread(p) for p in img_list]


def load_avatar(avatar_id):
    avatar_path = f"./data/avatars/{avatar_id}"
    bank_path = os.path.join(avatar_path, "bank.ltkbank")       # packed by livetalking_amd.bank.pack_avatar_dir
    if os.path.exists(bank_path):
        from ..bank import load_bank
        return load_bank(bank_path).as_avatar()
    with open(f"{avatar_path}/coords.pkl", "rb") as f:
        coord_list_cycle = pickle.load(f)

    def numbered(d):
        files = glob.glob(os.path.join(d, "*.[jpJP][pnPN]*[gG]"))
        return sorted(files, key=lambda x: int(os.path.splitext(os.path.basename(x))[0]))

    frame_list_cycle = read_imgs(numbered(f"{avatar_path}/full_imgs"))
    face_list_cycle = read_imgs(numbered(f"{avatar_path}/face_imgs"))
    return frame_list_cycle, face_list_cycle, coord_list_cycle


def warm_up(batch_size, model, modelres=256):
    """One forward on ones, as the reference does, to fault in kernels and arena."""
    mel = np.ones((batch_size, 80, 16), dtype=np.float32)
    img = np.ones((batch_size, 6, modelres, modelres), dtype=np.float32)
    for eng in model.engines:
        n = min(batch_size, eng.max_frames)
        eng.wav2lip_forward_host(mel[:n], img[:n])


@register("avatar", "wav2lip")
class LipReal(DeviceEgressMixin, BaseAvatar):
    _egress_source = SRC_WAV2LIP      # opt.egress = "bgr24" | "i420": device-side process_frames (egress.py)

    def __init__(self, opt, model, avatar):
        super().__init__(opt)
        self.model = model
        self.frame_list_cycle, self.face_list_cycle, self.coord_list_cycle = avatar
        self._slot = model.place(self)                  # this session's GPU for its whole life
        self.engine = model.engines[self._slot]
        self._aid = model.avatar_id(avatar, self._slot)
        h, w = self.frame_list_cycle[0].shape[:2]
        self._frame_hw = (int(h), int(w))
        self._sched = get_scheduler(self.engine)
        self.asr = MelASR(opt, self, engine=self.engine)
        self.asr.warm_up()

    def _mel_to_device(self, audiofeat_batch):
        import torch
        if isinstance(audiofeat_batch, torch.Tensor):
            return audiofeat_batch
        arr = np.ascontiguousarray(np.asarray(audiofeat_batch), dtype=np.float32)   # list of (80,16)
        return torch.from_numpy(arr).to(self.engine.torch_device)

    def inference_batch(self, index, audiofeat_batch):
        """Returns batch_size device handles (uint8 [256][256][3] BGR), item i for
        bank index mirror_index(len, index+i)."""
        import torch
        mel = self._mel_to_device(audiofeat_batch)
        B = self.batch_size
        if mel.shape[0] != B:
            raise ValueError(f"expected {B} mel windows, got {mel.shape[0]}")
        pred = torch.empty((B, 256, 256, 3), dtype=torch.uint8, device=mel.device)
        self._sched.infer(self._aid, int(index), B, mel.data_ptr(), pred.data_ptr())
        self._last_mel = mel            # keep inputs alive until the call returned (it has)
        # (everything from here to the next engine call is time the GPU idles when this is the only session: one unbind instead of
        # B slicings, bank indices left to the consumer's thread - 50 -> 20 us of Python per call, scripts/host_overhead.py)
        items = list(pred.unbind(0))
        if _PASTE_BATCH and hasattr(self.engine, "paste_back_batch"):
            # the process thread will ask for these B composites one by one, in order (base_avatar.py:429-433): remember
            # the batch so that the FIRST request composites all of them and moves them to the host in one copy
            FrameGroup.attach(items, pred, span=(len(self.frame_list_cycle), int(index), B))
        return items

    def paste_back_frame(self, pred_frame, idx: int):
        import torch
        if not isinstance(pred_frame, torch.Tensor):   # a float frame from a foreign inference_batch
            pred_frame = torch.from_numpy(np.ascontiguousarray(pred_frame).astype(np.uint8)).to(
                self.engine.torch_device)
        h, w = self._frame_hw
        grp = getattr(pred_frame, "_ltk_group", None)
        if grp is not None and grp.idx[pred_frame._ltk_i] == int(idx):
            with grp.lock:
                if grp.host is None:
                    # one pinned host block per batch: n composites on the GPU, ONE device-to-host copy at the PCIe rate.
                    # The returned frames are views of it; the block lives as long as any of them does (numpy base ->
                    # tensor -> torch's caching pinned allocator), so a consumer may keep or draw on a frame for as long
                    # as it likes - the reference's contract (a writable C-contiguous array the caller owns).
                    host = torch.empty((len(grp.idx), h, w, 3), dtype=torch.uint8, pin_memory=True)
                    self.engine.paste_back_batch(self._aid, grp.idx, grp.pred.data_ptr(), host.data_ptr())
                    grp.host = host.numpy()
            return grp.host[pred_frame._ltk_i]
        out = np.empty((h, w, 3), dtype=np.uint8)
        self.engine.paste_back(self._aid, int(idx), pred_frame.data_ptr(), out)
        return out
