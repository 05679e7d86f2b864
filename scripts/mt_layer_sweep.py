#!/usr/bin/env python3
"""Time the MuseTalk linear (1x1) and 3x3 layer geometries through ltk_conv2d_f16 / ltk_conv2d_fp8.  GPU only."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from livetalking_amd.engine import Engine  # noqa: E402
from livetalking_amd.layout import empty_cb16  # noqa: E402

N = int(os.environ.get("SWEEP_FRAMES", "16"))
# name, H, W, Cin, Cout, k
LAYERS = [
    ("q 320 @1024", 32, 32, 320, 320, 1), ("ff1 320>2560 @1024", 32, 32, 320, 2560, 1), ("ff2 1280>320 @1024", 32, 32, 1280, 320, 1),
    ("q 640 @256", 16, 16, 640, 640, 1), ("ff1 640>5120 @256", 16, 16, 640, 5120, 1), ("ff2 2560>640 @256", 16, 16, 2560, 640, 1),
    ("q 1280 @64", 8, 8, 1280, 1280, 1), ("ff1 1280>10240 @64", 8, 8, 1280, 10240, 1), ("ff2 5120>1280 @64", 8, 8, 5120, 1280, 1),
    ("kv 384>320 @50", 50, 1, 384, 320, 1), ("vae q 512 @1024", 32, 32, 512, 512, 1),
    ("sc 640>320 @1024", 32, 32, 640, 320, 1), ("sc 1920>640 @256", 16, 16, 1920, 640, 1),
    ("3x3 320 @32", 32, 32, 320, 320, 3), ("3x3 640 @16", 16, 16, 640, 640, 3), ("3x3 1280 @8", 8, 8, 1280, 1280, 3),
    ("3x3 2560>1280 @8", 8, 8, 2560, 1280, 3), ("3x3 1920>640 @16", 16, 16, 1920, 640, 3), ("3x3 960>320 @32", 32, 32, 960, 320, 3),
    ("3x3 512 @32", 32, 32, 512, 512, 3), ("3x3 512 @64", 64, 64, 512, 512, 3), ("3x3 512 @128", 128, 128, 512, 512, 3),
    ("3x3 256 @128", 128, 128, 256, 256, 3), ("3x3 256 @256", 256, 256, 256, 256, 3), ("3x3 128 @256", 256, 256, 128, 128, 3),
]


def main():
    eng = Engine(0)
    rng = np.random.default_rng(0)
    tot = {"f16": 0.0, "fp8": 0.0}
    for name, H, W, Cin, Cout, k in LAYERS:
        w = (rng.standard_normal((Cout, Cin, k, k)) * (2.0 / (Cin * k * k)) ** 0.5).astype(np.float32)
        x = torch.randn(N, Cin // 16, H, W, 16, device="cuda").half()
        y = empty_cb16(N, Cout, H, W)
        ms = eng.conv2d_f16(x.data_ptr(), N, H, W, Cin, w, Cout, k, 1, k // 2, False, 0, None, None, 0, False, y.data_ptr(), iters=5)
        flops = 2.0 * N * H * W * Cin * Cout * k * k
        line = f"{name:22s} f16 {ms*1e3:8.1f} us {flops/ms/1e9:7.1f} TF/s"
        if k == 3 and Cin % 32 == 0:
            xq = torch.randint(0, 120, (N, Cin // 32, H, W, 32), device="cuda", dtype=torch.uint8)
            ms8 = eng.conv2d_fp8(xq.data_ptr(), N, H, W, Cin, w, Cout, None, None, 8.0, 0, 0, y.data_ptr(), iters=5)
            line += f"   fp8 {ms8*1e3:8.1f} us {flops/ms8/1e9:7.1f} TF/s  x{ms/ms8:.2f}"
        print(line, flush=True)
    eng.close()


if __name__ == "__main__":
    main()
