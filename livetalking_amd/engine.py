"""Python handle on one per-GPU render engine (libltk_hip.so).

PyTorch-ROCm is used only as plumbing here: device buffers (`torch.empty(...,
device="cuda")`) and their `data_ptr()`s.  All arithmetic runs in the HIP
library.  One `Engine` per GPU; sessions are sharded across engines by the
host - there is no cross-GPU traffic.
"""
from __future__ import annotations

import ctypes as C
import threading
from typing import Dict, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import LtkError, MtReq, NamedTensor, W2lReq  # noqa: F401


def _as_f32(a) -> np.ndarray:
    if hasattr(a, "detach"):  # torch tensor
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(a, dtype=np.float32)


class Engine:
    """avatars/wav2lip_avatar.py `model` handle + device-resident avatar banks."""

    def __init__(self, device: int = 0):
        self._lib = _lib.load()
        h = C.c_void_p()
        _lib.check(self._lib.ltk_engine_create(int(device), C.byref(h)))
        self._h = h
        self.device = int(device)
        self.max_frames = 0
        self._closed = False
        self._lock = threading.Lock()
        self._egress_open = set()       # egress sessions still open: closed with the engine

    # ------------------------------------------------------------------ lifecycle
    def close(self):
        """Destroys the engine.  The handle is dropped with it: a later call on this object (a session's finalizer, a
        straggling paste) reaches the C ABI with a null engine and gets LTK_E_INVALID instead of touching freed memory."""
        if not self._closed:
            self._closed = True
            for sched in list(self.__dict__.get("_ltk_schedulers", {}).values()):     # scheduler.get_scheduler: stop the worker threads
                try:
                    sched.close()
                except Exception:
                    pass
            self.__dict__.pop("_ltk_schedulers", None)
            for sess in list(self._egress_open):
                try:
                    self.egress_close(sess)
                except Exception:
                    pass
            h, self._h = self._h, None
            self._lib.ltk_engine_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        _lib.check(self._lib.ltk_engine_sync(self._h))

    @property
    def torch_device(self):
        """Where the plugin allocates the device buffers it hands to this engine."""
        import torch
        return torch.device("cuda", self.device)

    # ------------------------------------------------------------------ model
    @staticmethod
    def _named_tensors(state_dict: Dict[str, object]):
        keep = []
        arr = (NamedTensor * len(state_dict))()
        n = 0
        for name, t in state_dict.items():
            if name.endswith("num_batches_tracked"):
                continue
            a = _as_f32(t)
            shape = (C.c_int64 * max(a.ndim, 1))(*a.shape)
            nm = name.encode()
            keep.append((a, shape, nm))
            arr[n].name = nm
            arr[n].data = a.ctypes.data_as(C.POINTER(C.c_float))
            arr[n].ndim = a.ndim
            arr[n].shape = shape
            n += 1
        return arr, n, keep

    def load_wav2lip(self, state_dict: Dict[str, object], max_frames: int = 16):
        """wav2lip_avatar.py:59-70: `state_dict` = checkpoint["state_dict"] with any
        "module." prefix stripped (tensors or arrays)."""
        arr, n, keep = self._named_tensors(state_dict)
        _lib.check(self._lib.ltk_wav2lip_load(self._h, arr, n, int(max_frames)))
        self.max_frames = int(max_frames)
        del keep

    def load_musetalk(self, unet_sd: Dict[str, object], vae_sd: Dict[str, object], max_frames: int = 16, fp8: bool = False,
                      fp8_act_scale: float = 0.0):
        """musetalk_avatar.py:57-67: diffusers-named state dicts of the U-Net (models/musetalkV15/unet.pth) and of
        the sd-vae AutoencoderKL (post_quant_conv.*, decoder.*).  fp8: ResnetBlock2D 3x3 convs on e4m3 operands
        (BASELINE configs[4])."""
        if fp8:
            _lib.check(self._lib.ltk_musetalk_set_fp8(self._h, 1, float(fp8_act_scale)))
        ua, un, k1 = self._named_tensors(unet_sd)
        va, vn, k2 = self._named_tensors(vae_sd)
        _lib.check(self._lib.ltk_musetalk_load(self._h, ua, un, va, vn, int(max_frames)))
        self.mt_max_frames = int(max_frames)
        del k1, k2

    # ------------------------------------------------------------------ avatars
    def register_avatar(self, face_list: Sequence[np.ndarray], frame_list: Sequence[np.ndarray],
                        coord_list: Sequence[Sequence[int]]) -> int:
        """Upload (frame_list_cycle, face_list_cycle, coord_list_cycle) as
        load_avatar returns them (wav2lip_avatar.py:72-88)."""
        # a packed bank (livetalking_amd/bank.py) hands over its contiguous mmap sections: no re-stacking, one H2D copy each
        faces = getattr(face_list, "packed", None)
        fulls = getattr(frame_list, "packed", None)
        faces = np.ascontiguousarray(np.stack(face_list) if faces is None else faces, dtype=np.uint8)
        fulls = np.ascontiguousarray(np.stack(frame_list) if fulls is None else fulls, dtype=np.uint8)
        coords = np.ascontiguousarray(np.asarray(coord_list, dtype=np.int32).reshape(-1, 4))
        n = faces.shape[0]
        if faces.shape[1:] != (256, 256, 3) or fulls.shape[0] != n or coords.shape[0] != n or fulls.shape[3] != 3:
            raise ValueError("avatar bank shapes: faces (n,256,256,3), frames (n,H,W,3), coords (n,4)")
        aid = C.c_int()
        _lib.check(self._lib.ltk_avatar_register(self._h, faces.ctypes.data, fulls.ctypes.data, coords.ctypes.data,
                                                 n, fulls.shape[1], fulls.shape[2], C.byref(aid)))
        return aid.value

    def release_avatar(self, avatar_id: int):
        _lib.check(self._lib.ltk_avatar_release(self._h, int(avatar_id)))

    # ------------------------------------------------------------------ hot path
    def mel_step(self, pcm: np.ndarray, win_start: Sequence[int], d_out_ptr: int, stream: int = 0):
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        ws = np.ascontiguousarray(win_start, dtype=np.int32)
        _lib.check(self._lib.ltk_mel_step(self._h, pcm.ctypes.data, pcm.shape[0], ws.ctypes.data, ws.shape[0],
                                          C.c_void_p(d_out_ptr), C.c_void_p(stream)))

    def wav2lip_infer(self, reqs: Sequence[tuple], stream: int = 0):
        """reqs: (avatar_id, index, batch, d_mel_ptr, d_pred_ptr) per session; each
        session's uint8 frames [batch][256][256][3] land in its own d_pred."""
        arr = (W2lReq * len(reqs))()
        for i, (aid, index, batch, mel_ptr, pred_ptr) in enumerate(reqs):
            arr[i].avatar = int(aid)
            arr[i].index = int(index)
            arr[i].batch = int(batch)
            arr[i].d_mel = C.c_void_p(mel_ptr)
            arr[i].d_pred = C.c_void_p(pred_ptr)
        _lib.check(self._lib.ltk_wav2lip_infer(self._h, arr, len(reqs), C.c_void_p(stream)))

    def paste_back(self, avatar_id: int, idx: int, d_pred_ptr: int, out: np.ndarray, stream: int = 0):
        """out: C-contiguous uint8 (H,W,3) host array, filled in place."""
        _lib.check(self._lib.ltk_paste_back(self._h, int(avatar_id), int(idx), C.c_void_p(d_pred_ptr),
                                            out.ctypes.data, 0, C.c_void_p(stream)))

    def paste_back_batch(self, avatar_id: int, idx: Sequence[int], d_pred_ptr: int, out_ptr: int, stream: int = 0):
        """n = len(idx) composites (n contiguous device predictions) into host memory at out_ptr [n][H][W][3] (pinned for
        the PCIe rate): one device-to-host copy, one synchronisation (include/ltk.h: ltk_paste_back_batch)."""
        arr = np.ascontiguousarray(np.asarray(idx, dtype=np.int32))
        _lib.check(self._lib.ltk_paste_back_batch(self._h, int(avatar_id), arr.ctypes.data, C.c_void_p(d_pred_ptr), int(arr.size),
                                                  C.c_void_p(out_ptr), C.c_void_p(stream)))

    def paste_back_device(self, avatar_id: int, idx: int, d_pred_ptr: int, d_out_ptr: int, stream: int = 0):
        _lib.check(self._lib.ltk_paste_back(self._h, int(avatar_id), int(idx), C.c_void_p(d_pred_ptr),
                                            C.c_void_p(d_out_ptr), 1, C.c_void_p(stream)))

    # ------------------------------------------------------------------ musetalk
    def register_musetalk_avatar(self, latents: Sequence[np.ndarray], frame_list: Sequence[np.ndarray], coord_list,
                                 mask_list: Sequence[np.ndarray], mask_coords_list) -> int:
        """(frame_list_cycle, mask_list_cycle, coord_list_cycle, mask_coords_list_cycle, input_latent_list_cycle) as
        musetalk_avatar.load_avatar returns them (musetalk_avatar.py:69-91)."""
        lat = np.ascontiguousarray(np.concatenate([_as_f32(x).reshape(1, 8, 32, 32) for x in latents]), dtype=np.float32)
        fulls = getattr(frame_list, "packed", None)
        fulls = np.ascontiguousarray(np.stack(frame_list) if fulls is None else fulls, dtype=np.uint8)
        fb = np.ascontiguousarray(np.asarray(coord_list, dtype=np.int32).reshape(-1, 4))
        cb = np.ascontiguousarray(np.asarray(mask_coords_list, dtype=np.int32).reshape(-1, 4))
        n = lat.shape[0]
        flat = [np.ascontiguousarray(m, dtype=np.uint8).reshape(-1) for m in mask_list]
        offs = np.zeros(n + 1, dtype=np.int64)
        offs[1:] = np.cumsum([f.size for f in flat])
        masks = np.ascontiguousarray(np.concatenate(flat))
        aid = C.c_int()
        _lib.check(self._lib.ltk_musetalk_avatar_register(self._h, lat.ctypes.data, fulls.ctypes.data, fb.ctypes.data, cb.ctypes.data,
                                                          masks.ctypes.data, offs.ctypes.data, n, fulls.shape[1], fulls.shape[2],
                                                          C.byref(aid)))
        return aid.value

    def musetalk_infer(self, reqs: Sequence[tuple], stream: int = 0):
        """reqs: (avatar_id, index, batch, d_feat_ptr fp32 [batch][50][384], d_pred_ptr uint8 [batch][256][256][3])."""
        arr = (MtReq * len(reqs))()
        for i, (aid, index, batch, feat_ptr, pred_ptr) in enumerate(reqs):
            arr[i].avatar = int(aid); arr[i].index = int(index); arr[i].batch = int(batch)
            arr[i].d_feat = C.c_void_p(feat_ptr); arr[i].d_pred = C.c_void_p(pred_ptr)
        _lib.check(self._lib.ltk_musetalk_infer(self._h, arr, len(reqs), C.c_void_p(stream)))

    def paste_blend(self, avatar_id: int, idx: int, d_pred_ptr: int, out: np.ndarray, stream: int = 0):
        _lib.check(self._lib.ltk_paste_blend(self._h, int(avatar_id), int(idx), C.c_void_p(d_pred_ptr), out.ctypes.data, 0,
                                             C.c_void_p(stream)))

    # ---- frame egress (include/ltk.h: ltk_egress_*)
    def egress_open(self, H: int, W: int) -> int:
        h = C.c_void_p()
        _lib.check(self._lib.ltk_egress_open(self._h, int(H), int(W), C.byref(h)))
        with self._lock:
            self._egress_open.add(h.value)
        return h.value

    def egress_close(self, session: int) -> None:
        """No-op on a session this engine no longer holds (closed already, or closed with the engine)."""
        with self._lock:
            if not self._h or session not in self._egress_open:
                return
            self._egress_open.discard(session)
        _lib.check(self._lib.ltk_egress_close(self._h, C.c_void_p(session)))

    def egress_watermark(self, session: int, mask, x: int = 0, y: int = 0, color=(128, 128, 128)) -> None:
        if mask is None:
            _lib.check(self._lib.ltk_egress_watermark(self._h, C.c_void_p(session), None, 0, 0, 0, 0, 0, 0, 0))
            return
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        _lib.check(self._lib.ltk_egress_watermark(self._h, C.c_void_p(session), mask.ctypes.data, int(x), int(y), mask.shape[1],
                                                  mask.shape[0], int(color[0]), int(color[1]), int(color[2])))

    def egress_frame(self, session: int, out: np.ndarray, source: int, avatar_id: int = 0, idx: int = 0, d_pred_ptr: int = 0,
                     h_frame: np.ndarray = None, speaking: bool = True, alpha: float = -1.0, keep: bool = False, fmt: int = 0,
                     chroma: int = 1, stream: int = 0) -> np.ndarray:
        req = _lib.EgressReq(int(source), int(avatar_id), int(idx), C.c_void_p(d_pred_ptr or None),
                             C.c_void_p(h_frame.ctypes.data if h_frame is not None else None), int(bool(speaking)), float(alpha),
                             int(bool(keep)), int(fmt), int(chroma))
        _lib.check(self._lib.ltk_egress_frame(self._h, C.c_void_p(session), C.byref(req), out.ctypes.data, C.c_void_p(stream)))
        return out

    def egress_batch(self, session: int, source: int, avatar_id: int, idx: Sequence[int], d_pred_ptr: int, out_ptr: int, fmt: int = 0,
                     chroma: int = 1, stream: int = 0) -> None:
        """The speaking frames of one inference_batch result: len(idx) composites + watermark + format conversion on the device, one
        device-to-host copy into host memory at out_ptr (include/ltk.h: ltk_egress_batch)."""
        arr = np.ascontiguousarray(np.asarray(idx, dtype=np.int32))
        _lib.check(self._lib.ltk_egress_batch(self._h, C.c_void_p(session), int(source), int(avatar_id), arr.ctypes.data,
                                              C.c_void_p(d_pred_ptr), int(arr.size), int(fmt), int(chroma), C.c_void_p(out_ptr),
                                              C.c_void_p(stream)))

    def musetalk_forward_host(self, latents: np.ndarray, feat: np.ndarray, want_image=True, want_frames=True):
        latents = np.ascontiguousarray(latents, dtype=np.float32).reshape(-1, 8, 32, 32)
        feat = np.ascontiguousarray(feat, dtype=np.float32).reshape(-1, 50, 384)
        B = latents.shape[0]
        unet_out = np.empty((B, 4, 32, 32), dtype=np.float32)
        image = np.empty((B, 3, 256, 256), dtype=np.float32) if want_image else None
        frames = np.empty((B, 256, 256, 3), dtype=np.uint8) if want_frames else None
        _lib.check(self._lib.ltk_musetalk_forward_host(self._h, latents.ctypes.data, feat.ctypes.data, B, unet_out.ctypes.data,
                                                       image.ctypes.data if want_image else None,
                                                       frames.ctypes.data if want_frames else None))
        return unet_out, image, frames

    def musetalk_debug_get(self, name: str, shape) -> np.ndarray:
        out = np.empty(shape, dtype=np.float32)
        _lib.check(self._lib.ltk_musetalk_debug_get(self._h, name.encode(), int(shape[0]), out.ctypes.data, out.size))
        return out

    def musetalk_info(self):
        macs, macs8 = C.c_double(), C.c_double()
        _lib.check(self._lib.ltk_musetalk_info(self._h, C.byref(macs), C.byref(macs8)))
        return macs.value, macs8.value

    def musetalk_ops(self):
        """[(name, type)] of the MuseTalk launch program; type 0 conv/linear, 1 GroupNorm, 2 LayerNorm, 3 attention, 4 GEGLU, 5 add-pos, 6 value transpose."""
        out = []
        for i in range(self._lib.ltk_musetalk_op_count(self._h)):
            buf, t = C.create_string_buffer(160), C.c_int()
            _lib.check(self._lib.ltk_musetalk_op_name(self._h, i, buf, 160, C.byref(t)))
            out.append((buf.value.decode(), t.value))
        return out

    def musetalk_time_ops(self, frames: int, iters: int) -> np.ndarray:
        n = self._lib.ltk_musetalk_op_count(self._h)
        ms = (C.c_float * n)()
        _lib.check(self._lib.ltk_musetalk_time_ops(self._h, int(frames), int(iters), ms, n))
        return np.array(ms[:], dtype=np.float64)

    def musetalk_time(self, frames: int, iters: int):
        ms = C.c_float()
        macs = C.c_double()
        _lib.check(self._lib.ltk_musetalk_time(self._h, int(frames), int(iters), C.byref(ms), C.byref(macs)))
        return ms.value, macs.value

    # ------------------------------------------------------------------ avatar preparation (VAE encoder)
    def load_vae_encoder(self, vae_sd: Dict[str, object], max_faces: int = 8):
        sd = {k: v for k, v in vae_sd.items() if k.startswith("encoder.") or k.startswith("quant_conv.")}
        arr, n, keep = self._named_tensors(sd)
        _lib.check(self._lib.ltk_vae_encoder_load(self._h, arr, n, int(max_faces)))
        del keep

    def vae_encode_faces(self, faces_bgr: np.ndarray, noise: Optional[np.ndarray] = None) -> np.ndarray:
        """vae.get_latents_for_unet for each 256x256 BGR crop -> fp32 (n, 8, 32, 32)."""
        faces = np.ascontiguousarray(faces_bgr, dtype=np.uint8).reshape(-1, 256, 256, 3)
        n = faces.shape[0]
        out = np.empty((n, 8, 32, 32), dtype=np.float32)
        nz = None if noise is None else np.ascontiguousarray(noise, dtype=np.float32).reshape(n, 2, 4, 32, 32)
        _lib.check(self._lib.ltk_vae_encode_faces(self._h, faces.ctypes.data, n, nz.ctypes.data if nz is not None else None,
                                                  out.ctypes.data))
        return out

    # ------------------------------------------------------------------ whisper audio features
    def load_whisper(self, encoder_sd: Dict[str, object]):
        """audio2feature.py:15-23: `encoder_sd` = WhisperModel.from_pretrained(...).encoder.state_dict()."""
        arr, n, keep = self._named_tensors(encoder_sd)
        _lib.check(self._lib.ltk_whisper_load(self._h, arr, n))
        del keep

    def whisper_step(self, pcm: np.ndarray, batch: int, first_row: int, d_out_ptr: int, row_step: int = 2, rows: int = 10,
                     stream: int = 0):
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        _lib.check(self._lib.ltk_whisper_step(self._h, pcm.ctypes.data, pcm.shape[0], int(batch), int(first_row), int(row_step),
                                              int(rows), C.c_void_p(d_out_ptr), C.c_void_p(stream)))

    def whisper_debug_get(self, name: str, shape) -> np.ndarray:
        out = np.empty(shape, dtype=np.float32)
        _lib.check(self._lib.ltk_whisper_debug_get(self._h, name.encode(), out.ctypes.data, out.size))
        return out

    # ------------------------------------------------------------------ test / measurement hooks
    def wav2lip_forward_host(self, mel: np.ndarray, face6: np.ndarray) -> np.ndarray:
        mel = np.ascontiguousarray(mel, dtype=np.float32).reshape(-1, 80, 16)
        face6 = np.ascontiguousarray(face6, dtype=np.float32)
        B = face6.shape[0]
        assert face6.shape[1:] == (6, 256, 256) and mel.shape[0] == B
        pred = np.empty((B, 3, 256, 256), dtype=np.float32)
        _lib.check(self._lib.ltk_wav2lip_forward_host(self._h, mel.ctypes.data, face6.ctypes.data, B, pred.ctypes.data))
        return pred

    @staticmethod
    def set_knob(name: str, value: int):
        """Process-wide tuning knob (csrc/tune.h); sweeps and A/B tests only."""
        lib = _lib.load()
        _lib.check(lib.ltk_debug_set_knob(name.encode(), int(value)))

    def saturation(self, reset: bool = True):
        """(values at the fp16 / e4m3 limit, non-finite values) seen in layer / op outputs since the last reset; counted only while
        knob SAT_CHECK is on (include/ltk.h: ltk_debug_saturation)."""
        a, b = C.c_ulonglong(), C.c_ulonglong()
        _lib.check(self._lib.ltk_debug_saturation(self._h, 1 if reset else 0, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def debug_capture(self, enable: bool):
        _lib.check(self._lib.ltk_debug_capture(self._h, 1 if enable else 0))

    def debug_get(self, layer: str, shape) -> np.ndarray:
        out = np.empty(shape, dtype=np.float32)
        _lib.check(self._lib.ltk_debug_get(self._h, layer.encode(), out.ctypes.data, out.size))
        return out

    def layer_names(self):
        n = self._lib.ltk_wav2lip_layer_count(self._h)
        out = []
        for i in range(n):
            buf = C.create_string_buffer(96)
            _lib.check(self._lib.ltk_wav2lip_layer_name(self._h, i, buf, 96))
            out.append(buf.value.decode())
        return out

    def set_layer_tile(self, layer: int, bucket: int, pxw: int = 0, nbt: int = 0, ksplit: int = 0):
        _lib.check(self._lib.ltk_wav2lip_set_layer_tile(self._h, int(layer), int(bucket), int(pxw), int(nbt), int(ksplit)))

    def time_layers(self, frames: int, iters: int) -> np.ndarray:
        n = self._lib.ltk_wav2lip_layer_count(self._h)
        ms = (C.c_float * n)()
        _lib.check(self._lib.ltk_wav2lip_time_layers(self._h, int(frames), int(iters), ms, n))
        return np.array(ms[:], dtype=np.float64)

    def graph_count(self) -> int:
        """Frame counts whose Wav2Lip pass currently replays from a captured hipGraph (include/ltk.h)."""
        return int(self._lib.ltk_wav2lip_graph_count(self._h))

    def face_cache_bytes(self, avatar_id: int) -> int:
        """Bytes of face-encoder skip cache the avatar holds (knob FACE_CACHE, include/ltk.h); 0 = none built."""
        n = C.c_size_t(0)
        _lib.check(self._lib.ltk_avatar_face_cache_bytes(self._h, int(avatar_id), C.byref(n)))
        return int(n.value)

    def prefetch_stats(self) -> dict:
        """Knob PREFETCH (include/ltk.h): single-request calls that found their face-encoder outputs prefetched by the previous
        call / that did not / prefetches issued."""
        h, m, i = C.c_ulonglong(0), C.c_ulonglong(0), C.c_ulonglong(0)
        _lib.check(self._lib.ltk_wav2lip_prefetch_stats(self._h, C.byref(h), C.byref(m), C.byref(i)))
        return {"hits": int(h.value), "misses": int(m.value), "issued": int(i.value)}

    def program_graph_count(self) -> int:
        """(program, frame count) pairs of the MuseTalk side (U-Net + VAE pass, Whisper encoder) that replay from a captured hipGraph."""
        return int(self._lib.ltk_program_graph_count(self._h))

    def time_convs(self, frames: int, iters: int):
        ms = C.c_float()
        macs = C.c_double()
        _lib.check(self._lib.ltk_wav2lip_time_convs(self._h, int(frames), int(iters), C.byref(ms), C.byref(macs)))
        return ms.value, macs.value

    def conv2d_fp8(self, d_x_ptr: int, N, H, W, Cin, weight: np.ndarray, Cout, scale=None, shift=None, act_scale: float = 8.0,
                   d_res_ptr: int = 0, act: int = 0, d_y_ptr: int = 0, iters: int = 0) -> float:
        """3x3 s1 p1 conv on e4m3 operands (include/ltk.h: ltk_conv2d_fp8); x is [N][Cin/32][H][W][32] bytes."""
        w = np.ascontiguousarray(weight, dtype=np.float32)
        sc = np.ascontiguousarray(scale, dtype=np.float32) if scale is not None else None
        sf = np.ascontiguousarray(shift, dtype=np.float32) if shift is not None else None
        ms = C.c_float(0)
        _lib.check(self._lib.ltk_conv2d_fp8(
            self._h, C.c_void_p(d_x_ptr), N, H, W, Cin, w.ctypes.data, Cout, sc.ctypes.data if sc is not None else None,
            sf.ctypes.data if sf is not None else None, float(act_scale), C.c_void_p(d_res_ptr) if d_res_ptr else None, int(act),
            C.c_void_p(d_y_ptr), iters, C.byref(ms)))
        return ms.value

    def groupnorm_f16(self, d_x_ptr: int, N, C_, P, groups, eps, gamma: np.ndarray, beta: np.ndarray, silu: bool, d_y_ptr: int, impl: int = 0,
                      out_fp8: bool = False, out_scale: float = 1.0, iters: int = 0) -> float:
        """Standalone GroupNorm(+SiLU) on a CB16 tensor (include/ltk.h: ltk_groupnorm_f16); impl 0 = the program's choice, 1 = two-pass,
        2 = one block per (image, group), 3 = one tensor pass with the exchange between blocks.  Returns ms per run when iters > 0."""
        ga = np.ascontiguousarray(gamma, dtype=np.float32)
        be = np.ascontiguousarray(beta, dtype=np.float32)
        ms = C.c_float(0)
        _lib.check(self._lib.ltk_groupnorm_f16(self._h, C.c_void_p(d_x_ptr), N, C_, P, groups, float(eps), ga.ctypes.data, be.ctypes.data,
                                               1 if silu else 0, impl, 1 if out_fp8 else 0, float(out_scale), C.c_void_p(d_y_ptr), iters, C.byref(ms)))
        return ms.value

    def conv2d_f16(self, d_x_ptr: int, N, H, W, Cin, weight: np.ndarray, Cout, k, stride, pad, transposed=False,
                   out_pad=0, scale: Optional[np.ndarray] = None, shift: Optional[np.ndarray] = None,
                   d_res_ptr: int = 0, relu=True, d_y_ptr: int = 0, iters: int = 0) -> float:
        w = np.ascontiguousarray(weight, dtype=np.float32)
        sc = np.ascontiguousarray(scale, dtype=np.float32) if scale is not None else None
        sf = np.ascontiguousarray(shift, dtype=np.float32) if shift is not None else None
        ms = C.c_float(0)
        kh, kw = (k, k) if isinstance(k, int) else k
        sh, sw = (stride, stride) if isinstance(stride, int) else stride
        ph, pw = (pad, pad) if isinstance(pad, int) else pad
        _lib.check(self._lib.ltk_conv2d_f16(
            self._h, C.c_void_p(d_x_ptr), N, H, W, Cin, w.ctypes.data, Cout, kh, kw, sh, sw, ph, pw,
            1 if transposed else 0, out_pad, sc.ctypes.data if sc is not None else None,
            sf.ctypes.data if sf is not None else None, C.c_void_p(d_res_ptr) if d_res_ptr else None,
            1 if relu else 0, C.c_void_p(d_y_ptr), iters, C.byref(ms)))
        return ms.value
