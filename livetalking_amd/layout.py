"""Activation layout of the HIP engine: channel-blocked "CB16" = [N][C/16][H][W][16] fp16.

One MFMA k-step (16 channels) is one channel block; patch rows, output rows and residual rows are
contiguous 32-byte-per-pixel runs (DESIGN.md "Data layout in HBM").  Tensors of <= 8 channels that enter
the network (packed face crops, mel windows) are plain [N][H][W][8].  These helpers convert torch NCHW
tensors for tests and tools; the engine itself never needs them.
"""
from __future__ import annotations


def to_cb16(x_nchw):
    """NCHW (any float dtype) -> fp16 [N][ceil(C/16)][H][W][16] (zero padded); C <= 8 -> [N][H][W][8]."""
    import torch
    n, c, h, w = x_nchw.shape
    if c <= 8:
        out = torch.zeros(n, h, w, 8, dtype=torch.float16, device=x_nchw.device)
        out[..., :c] = x_nchw.permute(0, 2, 3, 1).half()
        return out
    cb = (c + 15) // 16
    buf = torch.zeros(n, cb * 16, h, w, dtype=torch.float16, device=x_nchw.device)
    buf[:, :c] = x_nchw.half()
    return buf.view(n, cb, 16, h, w).permute(0, 1, 3, 4, 2).contiguous()


def from_cb16(y, channels: int):
    """[N][CB][H][W][16] -> NCHW float32 with the first `channels` channels."""
    n, cb, h, w, _ = y.shape
    return y.permute(0, 1, 4, 2, 3).reshape(n, cb * 16, h, w)[:, :channels].float()


def empty_cb16(n, channels, h, w, device="cuda", fill=None):
    import torch
    cb = (channels + 15) // 16
    if fill is None:
        return torch.empty(n, cb, h, w, 16, dtype=torch.float16, device=device)
    return torch.full((n, cb, h, w, 16), fill, dtype=torch.float16, device=device)


def to_cb32_fp8(x_nchw, scale: float = 1.0):
    """NCHW float -> e4m3 bytes [N][C/32][H][W][32] of saturate(x * scale) (C % 32 == 0): the operand layout of the fp8
    conv path.  Returns (uint8 tensor, the dequantised NCHW float32 tensor = what the kernel multiplies, / scale)."""
    import torch
    n, c, h, w = x_nchw.shape
    assert c % 32 == 0
    q = (x_nchw.float() * scale).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    packed = q.view(torch.uint8).view(n, c // 32, 32, h, w).permute(0, 1, 3, 4, 2).contiguous()
    return packed, q.float() / scale
