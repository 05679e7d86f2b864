"""Packed avatar bank: one mmap-able, GPU-uploadable file per avatar.

The reference keeps an avatar as a directory of hundreds of PNGs plus pickles and re-decodes all of them with
threaded cv2.imread at start-up (utils/image.py:14-24, avatars/wav2lip_avatar.py:72-88,
avatars/musetalk_avatar.py:69-91; writers: avatars/wav2lip/genavatar.py:124-138, avatars/musetalk/genavatar.py:
134-156).  A `.ltkbank` file holds the same payload as raw, 4096-byte-aligned arrays:

    header   magic "LTKBANK1", version, kind (1 wav2lip / 2 musetalk), n, H, W, n_sections
    table    n_sections x { name[16], offset u64, nbytes u64, dtype u32, ndim u32, shape u64[4] }
    sections wav2lip : face u8[n][256][256][3] BGR, full u8[n][H][W][3] BGR, coords i32[n][4] (y1,y2,x1,x2)
             musetalk: latents f32[n][8][32][32], full u8[n][H][W][3], face_boxes i32[n][4] (x1,y1,x2,y2),
                       crop_boxes i32[n][4] (x_s,y_s,x_e,y_e), masks u8[sum h_i*w_i*3], mask_offsets i64[n+1]

`load_bank` maps the file copy-on-write: frames are writable numpy views (the reference watermarks cached frames in
place, avatars/base_avatar.py:417,449) but the file never changes, start-up does no decoding, and the engine uploads
each section with ONE host-to-device copy straight from the page cache.
"""
from __future__ import annotations

import glob
import os
import pickle
import struct
from typing import List, Sequence

import numpy as np

MAGIC = b"LTKBANK1"
VERSION = 1
KIND_WAV2LIP, KIND_MUSETALK = 1, 2
ALIGN = 4096
_DTYPES = {0: np.uint8, 1: np.int32, 2: np.float32, 3: np.int64}
_CODES = {np.dtype(v): k for k, v in _DTYPES.items()}
_HDR = struct.Struct("<8sIIIIII")              # magic, version, kind, n, H, W, n_sections
_SEC = struct.Struct("<16sQQII4Q")


class PackedList(list):
    """A list of per-frame array views that remembers the contiguous array they slice (`.packed`), so the engine can
    upload the bank without re-stacking it."""
    packed = None


def _views(arr: np.ndarray) -> PackedList:
    out = PackedList(arr[i] for i in range(arr.shape[0]))
    out.packed = arr
    return out


def _write(path: str, kind: int, n: int, H: int, W: int, sections: Sequence[tuple]) -> None:
    table_end = _HDR.size + _SEC.size * len(sections)
    off = (table_end + ALIGN - 1) // ALIGN * ALIGN
    entries, blobs = [], []
    for name, arr in sections:
        arr = np.ascontiguousarray(arr)
        shape = list(arr.shape) + [0] * (4 - arr.ndim)
        entries.append(_SEC.pack(name.encode().ljust(16, b"\0"), off, arr.nbytes, _CODES[arr.dtype], arr.ndim, *shape))
        blobs.append((off, arr))
        off = (off + arr.nbytes + ALIGN - 1) // ALIGN * ALIGN
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        f.write(_HDR.pack(MAGIC, VERSION, kind, n, H, W, len(sections)))
        for e in entries:
            f.write(e)
        for o, arr in blobs:
            f.seek(o)
            f.write(arr.tobytes())
        f.truncate(off)
    os.replace(tmp, path)


def write_wav2lip_bank(path: str, frame_list, face_list, coord_list) -> None:
    """(frame_list_cycle, face_list_cycle, coord_list_cycle) as wav2lip_avatar.load_avatar returns them."""
    full = np.stack([np.asarray(f, dtype=np.uint8) for f in frame_list])
    face = np.stack([np.asarray(f, dtype=np.uint8) for f in face_list])
    coords = np.asarray(coord_list, dtype=np.int32).reshape(-1, 4)
    n = full.shape[0]
    if face.shape != (n, 256, 256, 3) or coords.shape[0] != n or full.ndim != 4 or full.shape[3] != 3:
        raise ValueError("wav2lip bank: faces (n,256,256,3), frames (n,H,W,3), coords (n,4)")
    _write(path, KIND_WAV2LIP, n, full.shape[1], full.shape[2], [("face", face), ("full", full), ("coords", coords)])


def write_musetalk_bank(path: str, frame_list, mask_list, coord_list, mask_coords_list, latent_list) -> None:
    """The 5-tuple musetalk_avatar.load_avatar returns (frames, masks, coords, mask_coords, latents)."""
    full = np.stack([np.asarray(f, dtype=np.uint8) for f in frame_list])
    n = full.shape[0]
    lat = np.concatenate([np.asarray(getattr(x, "numpy", lambda: x)(), dtype=np.float32).reshape(1, 8, 32, 32) for x in latent_list])
    fb = np.asarray(coord_list, dtype=np.int32).reshape(-1, 4)
    cb = np.asarray(mask_coords_list, dtype=np.int32).reshape(-1, 4)
    flat = [np.ascontiguousarray(m, dtype=np.uint8).reshape(-1) for m in mask_list]
    offs = np.zeros(n + 1, dtype=np.int64)
    offs[1:] = np.cumsum([f.size for f in flat])
    for i in range(n):
        if flat[i].size != (cb[i, 3] - cb[i, 1]) * (cb[i, 2] - cb[i, 0]) * 3:
            raise ValueError(f"mask {i} does not match its crop box")
    if lat.shape[0] != n or fb.shape[0] != n or cb.shape[0] != n:
        raise ValueError("musetalk bank: one latent / box / mask per frame")
    _write(path, KIND_MUSETALK, n, full.shape[1], full.shape[2],
           [("latents", lat), ("full", full), ("face_boxes", fb), ("crop_boxes", cb), ("masks", np.concatenate(flat)), ("mask_offsets", offs)])


class Bank:
    def __init__(self, path: str, writable: bool = True):
        self.path = path
        with open(path, "rb") as f:
            hdr = f.read(_HDR.size)
            magic, ver, self.kind, self.n, self.H, self.W, nsec = _HDR.unpack(hdr)
            if magic != MAGIC or ver != VERSION:
                raise ValueError(f"{path}: not a version-{VERSION} .ltkbank file")
            table = [_SEC.unpack(f.read(_SEC.size)) for _ in range(nsec)]
        mode = "c" if writable else "r"            # "c": copy-on-write, the file itself is never modified
        self.sections = {}
        for name, off, nbytes, code, ndim, *shape in table:
            if off % ALIGN:
                raise ValueError("unaligned section")
            shp = tuple(int(s) for s in shape[:ndim])
            self.sections[name.rstrip(b"\0").decode()] = np.memmap(path, dtype=_DTYPES[code], mode=mode, offset=off, shape=shp)

    def as_avatar(self):
        """The tuple the matching load_avatar returns; frame lists are PackedList views into the mapping."""
        s = self.sections
        if self.kind == KIND_WAV2LIP:
            return _views(s["full"]), _views(s["face"]), [tuple(int(v) for v in c) for c in s["coords"]]
        offs = s["mask_offsets"]
        cb = s["crop_boxes"]
        masks = PackedList()
        for i in range(self.n):
            h, w = int(cb[i, 3] - cb[i, 1]), int(cb[i, 2] - cb[i, 0])
            masks.append(s["masks"][int(offs[i]):int(offs[i + 1])].reshape(h, w, 3))
        masks.packed = s["masks"]
        lat = _views(s["latents"].reshape(self.n, 1, 8, 32, 32))
        return (_views(s["full"]), masks, [tuple(int(v) for v in c) for c in s["face_boxes"]],
                [tuple(int(v) for v in c) for c in cb], lat)


def load_bank(path: str, writable: bool = True) -> Bank:
    return Bank(path, writable)


# ---------------------------------------------------------------------------------------------- directory -> bank
def _numbered(d: str) -> List[str]:
    files = glob.glob(os.path.join(d, "*.[jpJP][pnPN]*[gG]"))
    return sorted(files, key=lambda x: int(os.path.splitext(os.path.basename(x))[0]))


def _imread_bgr(path: str) -> np.ndarray:
    """cv2.imread(path) semantics for 8-bit images (3-channel BGR) without OpenCV."""
    from PIL import Image
    with Image.open(path) as im:
        return np.ascontiguousarray(np.asarray(im.convert("RGB"))[..., ::-1])


def pack_avatar_dir(avatar_dir: str, out_path: str = None, kind: str = "wav2lip") -> str:
    """Convert the reference's on-disk avatar (full_imgs/, face_imgs/ or mask/ + latents.pt, *.pkl) into one bank."""
    out_path = out_path or os.path.join(avatar_dir, "bank.ltkbank")
    with open(os.path.join(avatar_dir, "coords.pkl"), "rb") as f:
        coords = pickle.load(f)
    frames = [_imread_bgr(p) for p in _numbered(os.path.join(avatar_dir, "full_imgs"))]
    if kind == "wav2lip":
        faces = [_imread_bgr(p) for p in _numbered(os.path.join(avatar_dir, "face_imgs"))]
        write_wav2lip_bank(out_path, frames, faces, coords)
    else:
        import torch
        latents = torch.load(os.path.join(avatar_dir, "latents.pt"), map_location="cpu")
        with open(os.path.join(avatar_dir, "mask_coords.pkl"), "rb") as f:
            mask_coords = pickle.load(f)
        masks = [_imread_bgr(p) for p in _numbered(os.path.join(avatar_dir, "mask"))]
        write_musetalk_bank(out_path, frames, masks, coords, mask_coords, latents)
    return out_path
