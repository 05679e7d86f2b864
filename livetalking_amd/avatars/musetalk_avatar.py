"""HIP-backed drop-in for avatars/musetalk_avatar.py.

Module contract the reference's app.py relies on (app.py:128-143,99):
  load_model() -> model handle                    (musetalk_avatar.py:57-67)
  load_avatar(avatar_id) -> avatar tuple          (musetalk_avatar.py:69-91)
  warm_up(batch_size, model)                      (musetalk_avatar.py:93-108)
  @register("avatar", "musetalk") class MuseReal(BaseAvatar) with
  inference_batch(index, audiofeat_batch) and paste_back_frame(pred_frame, idx)   (musetalk_avatar.py:110-164)

Underneath: the model handle owns a per-GPU Engine with the U-Net + VAE decoder program and the Whisper encoder;
the avatar's latents, frames and masks are uploaded to HBM once; `inference_batch` returns device handles (uint8
256x256x3 BGR crops, already rounded the way vae.decode_latents does); `paste_back_frame` blends on the GPU and
returns the writable C-contiguous uint8 (H,W,3) array the reference returns.
"""
from __future__ import annotations

import glob
import os
import pickle
import threading
import weakref

import numpy as np

from ..egress import SRC_MUSETALK, DeviceEgress, DeviceEgressMixin, FrameGroup
from ..engine import Engine
from ..hostshim import BaseAvatar, mirror_index, register  # noqa: F401  (mirror_index: part of the reference module's namespace)
from ..scheduler import get_scheduler
from ..sharding import EnginePool, visible_devices
from .audio_features.whisper import Audio2Feature, WhisperASR


_PASTE_BATCH = os.environ.get("LTK_PASTE_BATCH", "1") != "0"     # 0: one composite + one pageable copy per paste_back_frame call


class MuseTalkModel:
    """Opaque `model` object (the reference's tuple vae, unet, pe, timesteps, audio_processor).  One engine + Whisper
    encoder per GPU (sharding.EnginePool); a session is pinned to one of them when it is constructed."""

    def __init__(self, engine_or_pool, audio_processors):
        if isinstance(engine_or_pool, EnginePool):
            self.pool = engine_or_pool
        else:
            self.pool = EnginePool([getattr(engine_or_pool, "device", 0)], lambda d: engine_or_pool)
        self.engines = self.pool.engines
        self.audio_processors = audio_processors if isinstance(audio_processors, (list, tuple)) else [audio_processors]
        self._avatars = {}
        self._lock = threading.Lock()

    @property
    def engine(self) -> Engine:
        return self.engines[0]

    @property
    def audio_processor(self) -> Audio2Feature:
        return self.audio_processors[0]

    def place(self, session) -> int:
        slot = self.pool.place(id(session))
        weakref.finalize(session, self.pool.release, id(session))
        return slot

    def avatar_id(self, avatar, slot: int = 0) -> int:
        frame_list, mask_list, coord_list, mask_coords_list, latent_list = avatar
        key = (id(latent_list), slot)
        with self._lock:
            aid = self._avatars.get(key)
            if aid is None:
                aid = self.engines[slot].register_musetalk_avatar(latent_list, frame_list, coord_list, mask_list, mask_coords_list)
                self._avatars[key] = aid
            return aid


# AutoencoderKL checkpoints published before diffusers 0.13 (sd-vae-ft-mse among them) store the mid-block attention under
# the old AttentionBlock names; AutoencoderKL.from_pretrained renames them on load (diffusers
# ModelMixin._convert_deprecated_attention_blocks), which is why the reference (vae.py:24) never sees them.  The engine
# looks tensors up by the current names, so the same renaming happens here.
_DEPRECATED_ATTN = (("query", "to_q"), ("key", "to_k"), ("value", "to_v"), ("proj_attn", "to_out.0"))


def convert_deprecated_vae_attention(sd: dict) -> dict:
    out = {}
    for k, v in sd.items():
        nk = k
        if ".attentions." in k:
            for old, new in _DEPRECATED_ATTN:
                for leaf in ("weight", "bias"):
                    if k.endswith(f".{old}.{leaf}"):
                        nk = k[: -len(f"{old}.{leaf}")] + f"{new}.{leaf}"
            if nk.endswith(".weight") and getattr(v, "ndim", 0) == 4 and tuple(v.shape[2:]) == (1, 1) and \
                    any(nk.endswith(f".{n}.weight") for _, n in _DEPRECATED_ATTN):
                v = v.reshape(v.shape[0], v.shape[1])          # some exports keep the projections as 1x1 convs
        out[nk] = v
    return out


# The U-Net program of csrc/musetalk.hip is built for ONE topology: MuseTalk 1.5's `models/musetalkV15/musetalk.json`, which
# the reference hands to diffusers' UNet2DConditionModel(**config) (avatars/musetalk/utils/utils.py:15-31,
# avatars/musetalk/models/unet.py:36-46).  A deployment's file is read and every field that shapes the graph must agree;
# a different file is an error here, where the reference would build a different network.
UNET_TOPOLOGY = {
    "in_channels": 8, "out_channels": 4, "block_out_channels": [320, 640, 1280, 1280], "layers_per_block": 2,
    "down_block_types": ["CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"],
    "up_block_types": ["UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"],
    "cross_attention_dim": 384, "attention_head_dim": 8, "norm_num_groups": 32, "norm_eps": 1e-5, "act_fn": "silu",
    "flip_sin_to_cos": True, "freq_shift": 0, "downsample_padding": 1, "mid_block_scale_factor": 1,
    "center_input_sample": False,
}
# models/sd-vae/config.json (AutoencoderKL sd-vae-ft-mse): the fields the decoder program and decode_latents depend on
VAE_TOPOLOGY = {
    "latent_channels": 4, "out_channels": 3, "block_out_channels": [128, 256, 512, 512], "layers_per_block": 2,
    "norm_num_groups": 32, "act_fn": "silu", "scaling_factor": 0.18215,
    "up_block_types": ["UpDecoderBlock2D"] * 4,
}


def check_model_config(path: str, expected: dict, what: str) -> bool:
    """Reads a diffusers config JSON if it exists and compares the graph-shaping fields with the compiled topology.
    Returns False when there is no file (state dicts handed over directly), raises ValueError on a mismatch."""
    import json
    if not os.path.exists(path):
        return False
    with open(path) as f:
        cfg = json.load(f)
    bad = []
    for k, want in expected.items():
        if k not in cfg:
            continue                                    # diffusers fills absent fields with its defaults = these values
        got = cfg[k]
        if isinstance(want, float):
            ok = isinstance(got, (int, float)) and abs(float(got) - want) <= 1e-9 * max(1.0, abs(want))
        elif isinstance(want, list):
            ok = list(got) == want
        else:
            ok = got == want
        if not ok:
            bad.append(f"{k}: file has {got!r}, engine is built for {want!r}")
    if bad:
        raise ValueError(f"{path}: {what} topology differs from the one csrc/musetalk.hip implements: " + "; ".join(bad))
    return True


def _read_vae_checkpoint():
    import torch
    from safetensors.torch import load_file   # sd-vae ships diffusion_pytorch_model.safetensors / .bin
    p = os.path.join("models", "sd-vae", "diffusion_pytorch_model.safetensors")
    return load_file(p) if os.path.exists(p) else torch.load(p.replace(".safetensors", ".bin"), map_location="cpu")


def load_model(unet_state_dict=None, vae_state_dict=None, whisper_encoder_state_dict=None, max_frames=None, device=None):
    """The reference reads models/musetalkV15/unet.pth, models/sd-vae and models/whisper
    (avatars/musetalk/utils/utils.py:16-37, audio2feature.py:15-23); state dicts may be passed directly (tests and
    the bench use seeded synthetic weights: none of those checkpoints exists in the reference tree).
    One engine per GPU of LTK_DEVICES (default: every visible GPU), or just `device`."""
    import torch
    if unet_state_dict is None:
        check_model_config(os.path.join("models", "musetalkV15", "musetalk.json"), UNET_TOPOLOGY, "U-Net")
        unet_state_dict = torch.load(os.path.join("models", "musetalkV15", "unet.pth"), map_location="cpu")
    if vae_state_dict is None:
        check_model_config(os.path.join("models", "sd-vae", "config.json"), VAE_TOPOLOGY, "VAE")
        vae_state_dict = _read_vae_checkpoint()
    vae_state_dict = convert_deprecated_vae_attention(vae_state_dict)
    vae_state_dict = {k: v for k, v in vae_state_dict.items() if k.startswith("decoder.") or k.startswith("post_quant_conv.")}
    if whisper_encoder_state_dict is None:
        from transformers import WhisperModel   # the checkpoint reader the reference uses (audio2feature.py:20-23)
        from .audio_features.whisper import check_whisper_dir
        check_whisper_dir("./models/whisper")
        whisper_encoder_state_dict = WhisperModel.from_pretrained("./models/whisper").encoder.state_dict()
    if max_frames is None:
        max_frames = int(os.environ.get("LTK_MT_MAX_FRAMES", "64"))
    # LTK_MT_FP8=1: the fp8 conv path of BASELINE.json configs[4] (ResnetBlock2D convs on e4m3 operands, include/ltk.h
    # ltk_musetalk_set_fp8); LTK_MT_FP8_ASCALE overrides the activation scale (default 8)
    fp8 = os.environ.get("LTK_MT_FP8", "0") not in ("", "0")

    def factory(dev):
        eng = Engine(dev)
        eng.load_musetalk(unet_state_dict, vae_state_dict, max_frames=max_frames, fp8=fp8,
                          fp8_act_scale=float(os.environ.get("LTK_MT_FP8_ASCALE", "0")))
        return eng

    devices = visible_devices() if device is None else [int(device)]
    cap = int(os.environ.get("LTK_SESSIONS_PER_GPU", "0")) or (1 << 30)
    pool = EnginePool(devices, factory, capacity_per_gpu=cap)
    aps = [Audio2Feature(eng, whisper_encoder_state_dict) for eng in pool.engines]
    return MuseTalkModel(pool, aps)


def read_imgs(img_list):
    import cv2  # same third-party reader the reference uses (utils/image.py:14-24)
    return [cv2.imread(p) for p in img_list]


def load_avatar(avatar_id):
    import torch
    avatar_path = f"./data/avatars/{avatar_id}"
    bank_path = os.path.join(avatar_path, "bank.ltkbank")       # packed by livetalking_amd.bank.pack_avatar_dir
    if os.path.exists(bank_path):
        from ..bank import load_bank
        return load_bank(bank_path).as_avatar()

    def numbered(d):
        files = glob.glob(os.path.join(d, "*.[jpJP][pnPN]*[gG]"))
        return sorted(files, key=lambda x: int(os.path.splitext(os.path.basename(x))[0]))

    input_latent_list_cycle = torch.load(f"{avatar_path}/latents.pt")
    with open(f"{avatar_path}/coords.pkl", "rb") as f:
        coord_list_cycle = pickle.load(f)
    frame_list_cycle = read_imgs(numbered(f"{avatar_path}/full_imgs"))
    with open(f"{avatar_path}/mask_coords.pkl", "rb") as f:
        mask_coords_list_cycle = pickle.load(f)
    mask_list_cycle = read_imgs(numbered(f"{avatar_path}/mask"))
    return frame_list_cycle, mask_list_cycle, coord_list_cycle, mask_coords_list_cycle, input_latent_list_cycle


def warm_up(batch_size, model):
    """One forward on ones, as the reference does (musetalk_avatar.py:93-108)."""
    for eng in model.engines:
        n = min(batch_size, eng.mt_max_frames)
        eng.musetalk_forward_host(np.ones((n, 8, 32, 32), np.float32), np.ones((n, 50, 384), np.float32),
                                  want_image=False, want_frames=False)


@register("avatar", "musetalk")
class MuseReal(DeviceEgressMixin, BaseAvatar):
    _egress_source = SRC_MUSETALK     # opt.egress = "bgr24" | "i420": device-side process_frames (egress.py)

    def __init__(self, opt, model, avatar):
        super().__init__(opt)
        self.model = model
        (self.frame_list_cycle, self.mask_list_cycle, self.coord_list_cycle, self.mask_coords_list_cycle,
         self.input_latent_list_cycle) = avatar
        self._slot = model.place(self)                  # this session's GPU for its whole life
        self.engine = model.engines[self._slot]
        self._aid = model.avatar_id(avatar, self._slot)
        h, w = self.frame_list_cycle[0].shape[:2]
        self._frame_hw = (int(h), int(w))
        self._sched = get_scheduler(self.engine, "musetalk")
        self._paste_eg = None           # lazily: the BGR24 egress session behind the batched paste_back_frame
        self.asr = WhisperASR(opt, self, model.audio_processors[self._slot])
        self.asr.warm_up()

    def inference_batch(self, index, audiofeat_batch):
        """Returns batch_size device handles (uint8 [256][256][3] BGR), item i for bank index
        mirror_index(len, index+i)."""
        import torch
        dev = self.engine.torch_device
        if isinstance(audiofeat_batch, torch.Tensor):
            feat = audiofeat_batch.to(device=dev, dtype=torch.float32).contiguous()
        else:                                               # list of (50,384) arrays from a foreign ASR
            feat = torch.from_numpy(np.ascontiguousarray(np.stack(audiofeat_batch), dtype=np.float32)).to(dev)
        B = self.batch_size
        if feat.shape[0] != B:
            raise ValueError(f"expected {B} whisper chunks, got {feat.shape[0]}")
        pred = torch.empty((B, 256, 256, 3), dtype=torch.uint8, device=dev)
        self._sched.infer(self._aid, int(index), B, feat.data_ptr(), pred.data_ptr())
        items = list(pred.unbind(0))
        if hasattr(self.engine, "egress_batch"):        # opt.egress sessions convert the batch's frames in one go (egress.py)
            FrameGroup.attach(items, pred, span=(len(self.frame_list_cycle), int(index), B))
        return items

    def paste_back_frame(self, pred_frame, idx: int):
        import torch
        if not isinstance(pred_frame, torch.Tensor):
            pred_frame = torch.from_numpy(np.ascontiguousarray(pred_frame).astype(np.uint8)).to(
                self.engine.torch_device)
        h, w = self._frame_hw
        grp = getattr(pred_frame, "_ltk_group", None)
        if grp is not None and _PASTE_BATCH and grp.idx[pred_frame._ltk_i] == int(idx):
            # the B blended composites of the batch on the GPU and ONE pinned device-to-host copy: a watermark-less BGR24 egress
            # session delivers exactly paste_back_frame's array (ltk_egress_batch = ltk_paste_blend per frame + a plain copy-out)
            if self._paste_eg is None:
                self._paste_eg = DeviceEgress(self.engine, h, w, SRC_MUSETALK, self._aid, fmt="bgr24", watermark=None)
                weakref.finalize(self, self._paste_eg.close)
            return self._paste_eg.speaking_frame_of(pred_frame, int(idx))
        out = np.empty((h, w, 3), dtype=np.uint8)
        self.engine.paste_blend(self._aid, int(idx), pred_frame.data_ptr(), out)
        return out
