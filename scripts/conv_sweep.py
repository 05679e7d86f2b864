#!/usr/bin/env python3
"""Time the Wav2Lip layer geometries through ltk_conv2d_f16 under kernel-config
overrides (LTK_CONV_MODE / LTK_CONV_NBT / LTK_CONV_NC8).  GPU only."""
import itertools
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from livetalking_amd.engine import Engine  # noqa: E402

N = int(os.environ.get("SWEEP_FRAMES", "16"))
# name, H, W, Cin, Cout, k, stride, pad, transposed, out_pad, residual
LAYERS = [
    ("c64@256", 256, 256, 64, 64, 3, 1, 1, False, 0, True),
    ("c128@128", 128, 128, 128, 128, 3, 1, 1, False, 0, True),
    ("c256@64", 64, 64, 256, 256, 3, 1, 1, False, 0, True),
    ("c384@32", 32, 32, 384, 384, 3, 1, 1, False, 0, True),
    ("c512@16", 16, 16, 512, 512, 3, 1, 1, False, 0, True),
    ("c512@8", 8, 8, 512, 512, 3, 1, 1, False, 0, True),
    ("out80>32@256", 256, 256, 80, 32, 3, 1, 1, False, 0, False),
    ("T160>64@128", 128, 128, 160, 64, 3, 2, 1, True, 1, False),
    ("T320>128@64", 64, 64, 320, 128, 3, 2, 1, True, 1, False),
    ("T512>256@32", 32, 32, 512, 256, 3, 2, 1, True, 1, False),
    ("T768>384@16", 16, 16, 768, 384, 3, 2, 1, True, 1, False),
    ("c7x7 6>16@256", 256, 256, 6, 16, 7, 1, 3, False, 0, False),
    ("s2 16>32@256", 256, 256, 16, 32, 3, 2, 1, False, 0, False),
    ("c32@128", 128, 128, 32, 32, 3, 1, 1, False, 0, True),
    ("s2 32>64@128", 128, 128, 32, 64, 3, 2, 1, False, 0, False),
    ("s2 64>128@64", 64, 64, 64, 128, 3, 2, 1, False, 0, False),
    ("s2 128>256@32", 32, 32, 128, 256, 3, 2, 1, False, 0, False),
    ("s2 256>512@16", 16, 16, 256, 512, 3, 2, 1, False, 0, False),
    ("s2 512>512@8", 8, 8, 512, 512, 3, 2, 1, False, 0, False),
    ("s2 320>320@32", 32, 32, 320, 320, 3, 2, 1, False, 0, False),
    ("c256@16", 16, 16, 256, 256, 3, 1, 1, False, 0, True),
    ("c512@4", 4, 4, 512, 512, 3, 1, 1, False, 0, True),
    ("T1024>512@8", 8, 8, 1024, 512, 3, 2, 1, True, 1, False),
    ("T1024>512@4", 4, 4, 1024, 512, 3, 2, 1, True, 1, False),
    ("1x1 512@1", 1, 1, 512, 512, 1, 1, 0, False, 0, False),
    ("T4x4 1024>512@1", 1, 1, 1024, 512, 4, 1, 0, True, 0, False),
]


def macs(l):
    _, H, W, Cin, Cout, k, s, p, tr, op, res = l
    if tr:
        return Cin * Cout * k * k * H * W
    Ho = (H + 2 * p - k) // s + 1
    Wo = (W + 2 * p - k) // s + 1
    return Cin * Cout * k * k * Ho * Wo


def main():
    eng = Engine(0)
    # (LTK_CONV_V3, LTK_CONV_MODE, LTK_CONV_NBT, LTK_CONV_NC8)
    variants = [(0, 1, 0, 0), (1, 1, 0, 0)]
    if os.environ.get("SWEEP_V3TILES"):
        variants = [(1, 1, 0, 0), ("pxw2", 1, 0, 0), ("nbt1", 1, 0, 0), ("pxw2nbt1", 1, 0, 0)]
    if os.environ.get("SWEEP_FULL"):
        variants = [(0, m, n, c) for m, n, c in itertools.product((1, 0), (0, 2, 1), (0, 2))] + [(1, 1, 0, 0)]
    only = os.environ.get("SWEEP_ONLY")
    print(f"frames={N}   cell = us / TFLOP/s")
    hdr = "layer".ljust(16) + "".join(((str(v) if isinstance(v, str) else "v3" if v else f"old m{m}n{n}c{c}")).rjust(14) for v, m, n, c in variants)
    print(hdr)
    for l in LAYERS:
        name, H, W, Cin, Cout, k, s, p, tr, op, res = l
        if only and only not in name:
            continue
        cin_pad = 8 if Cin <= 8 else (Cin + 15) // 16 * 16     # layout does not matter for timing: same bytes
        x = (torch.randn(N, H, W, cin_pad, device="cuda") * 0.5).half()
        wshape = (Cin, Cout, k, k) if tr else (Cout, Cin, k, k)
        w = (np.random.default_rng(0).standard_normal(wshape) * 0.05).astype(np.float32)
        if tr and s == 2:
            Ho, Wo = H * 2, W * 2
        elif tr:
            Ho, Wo = k, k
        else:
            Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        y = torch.empty(N, Ho, Wo, (Cout + 15) // 16 * 16, dtype=torch.float16, device="cuda")
        sc = np.ones(Cout, np.float32)
        sf = np.zeros(Cout, np.float32)
        row = name.ljust(16)
        for v3, mode, nbt, nc8 in variants:
            Engine.set_knob("CONV_PXW", 0); Engine.set_knob("CONV3_NBT", 0)
            if isinstance(v3, str):
                if "pxw2" in v3: Engine.set_knob("CONV_PXW", 2)
                if "nbt1" in v3: Engine.set_knob("CONV3_NBT", 1)
                v3 = 1
            Engine.set_knob("CONV_V3", v3)
            Engine.set_knob("CONV_MODE", mode)
            Engine.set_knob("CONV_NBT", nbt)
            Engine.set_knob("CONV_NC8", nc8)
            try:
                ms = eng.conv2d_f16(x.data_ptr(), N, H, W, Cin, w, Cout, k, s, p, tr, op, sc, sf,
                                    x.data_ptr() if res else 0, True, y.data_ptr(), iters=10)
                tf = 2 * macs(l) * N / ms / 1e9
                row += f" {ms*1e3:6.0f}/{tf:3.0f}".rjust(14)
            except Exception as ex:
                row += " err".rjust(14)
                print("   !", name, ex)
        print(row, flush=True)
    eng.close()


if __name__ == "__main__":
    main()
