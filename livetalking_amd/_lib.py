"""ctypes binding of libltk_hip.so (include/ltk.h).

The HIP library IS the product path: importing this module without the built
shared object raises, there is no CPU/PyTorch fallback.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LTK_LIB") or os.path.join(_HERE, "libltk_hip.so")      # LTK_LIB: A/B runs against another build


class LtkError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"ltk error {code}: {msg}")
        self.code = code


class NamedTensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.POINTER(C.c_float)), ("ndim", C.c_int),
                ("shape", C.POINTER(C.c_int64))]


class W2lReq(C.Structure):
    _fields_ = [("avatar", C.c_int), ("index", C.c_int), ("batch", C.c_int), ("d_mel", C.c_void_p),
                ("d_pred", C.c_void_p)]


class MtReq(C.Structure):
    _fields_ = [("avatar", C.c_int), ("index", C.c_int), ("batch", C.c_int), ("d_feat", C.c_void_p), ("d_pred", C.c_void_p)]


class EgressReq(C.Structure):
    _fields_ = [("source", C.c_int), ("avatar", C.c_int), ("idx", C.c_int), ("d_pred", C.c_void_p), ("h_frame", C.c_void_p),
                ("speaking", C.c_int), ("alpha", C.c_double), ("keep", C.c_int), ("format", C.c_int), ("chroma", C.c_int)]


# every symbol include/ltk.h declares: (restype, argtypes)
SYMBOLS = {
    "ltk_last_error": (C.c_char_p, []),
    "ltk_version": (C.c_char_p, []),
    "ltk_engine_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "ltk_engine_destroy": (None, [C.c_void_p]),
    "ltk_engine_sync": (C.c_int, [C.c_void_p]),
    "ltk_wav2lip_load": (C.c_int, [C.c_void_p, C.POINTER(NamedTensor), C.c_int, C.c_int]),
    "ltk_avatar_register": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                      C.POINTER(C.c_int)]),
    "ltk_avatar_release": (C.c_int, [C.c_void_p, C.c_int]),
    "ltk_mel_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "ltk_wav2lip_infer": (C.c_int, [C.c_void_p, C.POINTER(W2lReq), C.c_int, C.c_void_p]),
    "ltk_paste_back": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "ltk_paste_back_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "ltk_musetalk_load": (C.c_int, [C.c_void_p, C.POINTER(NamedTensor), C.c_int, C.POINTER(NamedTensor), C.c_int, C.c_int]),
    "ltk_musetalk_set_fp8": (C.c_int, [C.c_void_p, C.c_int, C.c_float]),
    "ltk_musetalk_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "ltk_musetalk_avatar_register": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "ltk_musetalk_infer": (C.c_int, [C.c_void_p, C.POINTER(MtReq), C.c_int, C.c_void_p]),
    "ltk_paste_blend": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "ltk_egress_open": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "ltk_egress_close": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ltk_egress_watermark": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "ltk_egress_frame": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(EgressReq), C.c_void_p, C.c_void_p]),
    "ltk_egress_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                   C.c_void_p, C.c_void_p]),
    "ltk_musetalk_forward_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ltk_musetalk_debug_get": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_size_t]),
    "ltk_musetalk_time": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_double)]),
    "ltk_whisper_load": (C.c_int, [C.c_void_p, C.POINTER(NamedTensor), C.c_int]),
    "ltk_whisper_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "ltk_whisper_debug_get": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]),
    "ltk_vae_encoder_load": (C.c_int, [C.c_void_p, C.POINTER(NamedTensor), C.c_int, C.c_int]),
    "ltk_vae_encode_faces": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "ltk_wav2lip_forward_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "ltk_debug_capture": (C.c_int, [C.c_void_p, C.c_int]),
    "ltk_debug_saturation": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
    "ltk_debug_set_knob": (C.c_int, [C.c_char_p, C.c_int]),
    "ltk_debug_tile_table_check": (C.c_int, [C.c_char_p, C.c_int]),
    "ltk_debug_get": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]),
    "ltk_wav2lip_time_convs": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_double)]),
    "ltk_wav2lip_graph_count": (C.c_int, [C.c_void_p]),
    "ltk_program_graph_count": (C.c_int, [C.c_void_p]),
    "ltk_wav2lip_prefetch_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
    "ltk_avatar_face_cache_bytes": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_size_t)]),
    "ltk_musetalk_op_count": (C.c_int, [C.c_void_p]),
    "ltk_musetalk_op_name": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int)]),
    "ltk_musetalk_time_ops": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int]),
    "ltk_wav2lip_layer_count": (C.c_int, [C.c_void_p]),
    "ltk_wav2lip_layer_name": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int]),
    "ltk_wav2lip_set_layer_tile": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "ltk_wav2lip_time_layers": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int]),
    "ltk_conv2d_f16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                 C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                 C.POINTER(C.c_float)]),
    "ltk_groupnorm_f16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                    C.c_int, C.c_float, C.c_void_p, C.c_int, C.POINTER(C.c_float)]),
    "ltk_f32_to_e4m3": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p]),
    "ltk_conv2d_fp8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                 C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_float)]),
}

_lib = None


def load() -> C.CDLL:
    """Load the shared library (once) and bind every declared symbol."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C livetalking_amd/csrc`). livetalking_amd has no CPU fallback.")
    # PyTorch-ROCm first: it ships its own libamdhip64.so.7 / HSA runtime, the engine links against the system's by the same soname, and
    # whichever is mapped first serves both.  With the system's mapped first (this library loaded before `import torch`, e.g. build() and
    # smoke() in one process) the second runtime finds no device ("no ROCm-capable device is detected"); with torch's first everything -
    # torch's allocator and the engine's streams - shares one runtime.  The plugin needs torch for its device buffers anyway.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        msg = load().ltk_last_error()
        raise LtkError(rc, msg.decode("utf-8", "replace") if msg else "")
