#!/usr/bin/env python3
"""Ablation timing of conv3 on the big layers: LTK_ABLATE bitmask (1 no A DMA, 2 no B DMA, 4 no MFMA,
8 no residual read, 16 no output store, 32 no zero fill).  GPU only; results are NOT valid convolutions."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from livetalking_amd.engine import Engine  # noqa: E402

N = int(os.environ.get("SWEEP_FRAMES", "16"))
LAYERS = [
    ("c64@256", 256, 256, 64, 64, 3, 1, 1, False, 0, True),
    ("c128@128", 128, 128, 128, 128, 3, 1, 1, False, 0, True),
    ("c256@64", 64, 64, 256, 256, 3, 1, 1, False, 0, True),
    ("c384@32", 32, 32, 384, 384, 3, 1, 1, False, 0, True),
    ("out80>32@256", 256, 256, 80, 32, 3, 1, 1, False, 0, False),
    ("T160>64@128", 128, 128, 160, 64, 3, 2, 1, True, 1, False),
    ("T320>128@64", 64, 64, 320, 128, 3, 2, 1, True, 1, False),
    # the small-map, deep-K layers (weights dominate; ABLATE_SET=small)
    ("c512@16", 16, 16, 512, 512, 3, 1, 1, False, 0, True),
    ("c512@8", 8, 8, 512, 512, 3, 1, 1, False, 0, True),
    ("c512@4", 4, 4, 512, 512, 3, 1, 1, False, 0, True),
    ("T1024>512@8", 8, 8, 1024, 512, 3, 2, 1, True, 1, False),
    ("T1024>512@4", 4, 4, 1024, 512, 3, 2, 1, True, 1, False),
    ("T768>384@16", 16, 16, 768, 384, 3, 2, 1, True, 1, False),
    ("s2 256>512@16", 16, 16, 256, 512, 3, 2, 1, False, 0, False),
    ("s2 512>512@8", 8, 8, 512, 512, 3, 2, 1, False, 0, False),
]
_SET = os.environ.get("ABLATE_SET", "")
if _SET == "small":
    LAYERS = [l for l in LAYERS if l[0].endswith(("@16", "@8", "@4"))]
elif _SET == "big":
    LAYERS = [l for l in LAYERS if not l[0].endswith(("@16", "@8", "@4"))]
MASKS = [int(m) for m in os.environ.get("ABLATE_MASKS", "0,3,4,24,27,31,63,95,127,64,128,132").split(",")]


def main():
    eng = Engine(0)
    Engine.set_knob("CONV_V3", 1)          # needs the measurement build: make -C livetalking_amd/csrc clean all ABLATE=1
    print(f"frames={N}; us per launch by LTK_ABLATE mask")
    print("layer".ljust(16) + "".join(f"{m:>8d}" for m in MASKS))
    for (name, H, W, Cin, Cout, k, s, p, tr, op, res) in LAYERS:
        x = (torch.randn(N, H, W, Cin, device="cuda") * 0.5).half()
        wshape = (Cin, Cout, k, k) if tr else (Cout, Cin, k, k)
        w = (np.random.default_rng(0).standard_normal(wshape) * 0.05).astype(np.float32)
        Ho, Wo = (H * 2, W * 2) if tr else (H, W)
        y = torch.empty(N, Ho, Wo, (Cout + 15) // 16 * 16, dtype=torch.float16, device="cuda")
        sc = np.ones(Cout, np.float32)
        sf = np.zeros(Cout, np.float32)
        row = name.ljust(16)
        for m in MASKS:
            Engine.set_knob("ABLATE", m)
            ms = eng.conv2d_f16(x.data_ptr(), N, H, W, Cin, w, Cout, k, s, p, tr, op, sc, sf,
                                x.data_ptr() if res else 0, True, y.data_ptr(), iters=10)
            row += f"{ms*1e3:8.0f}"
        print(row, flush=True)
    Engine.set_knob("ABLATE", 0)
    eng.close()


if __name__ == "__main__":
    main()
