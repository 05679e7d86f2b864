#!/usr/bin/env python3
"""lin_fk_kernel against conv3's 1x1 path on the linear-layer shapes of MuseTalk's 32^2 / 16^2 transformer levels (16 frames):
single-layer launches through ltk_conv2d_f16, knob LIN_FK = 0 / 1 interleaved, average of `iters` back-to-back launches.  GPU only.

    python scripts/lin_fk_ab.py [iters] [LIN_FK_BLOCKS values...]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from livetalking_amd.engine import Engine  # noqa: E402
from livetalking_amd.layout import empty_cb16, to_cb16  # noqa: E402

SHAPES = [  # (N, H, Cin, Cout, residual, name)
    (16, 32, 320, 320, True, "32^2 to_out / to_q / proj"), (16, 32, 320, 960, False, "32^2 to_qkv"), (16, 32, 320, 2560, False, "32^2 ff.net.0.proj"),
    (16, 16, 640, 640, True, "16^2 to_out / to_q / proj"), (16, 16, 640, 1920, False, "16^2 to_qkv"), (16, 16, 640, 5120, False, "16^2 ff.net.0.proj"),
    (64, 32, 320, 320, True, "64 frames 32^2 to_out"), (64, 16, 640, 640, True, "64 frames 16^2 to_out"),
    (4, 32, 320, 320, True, "4 frames 32^2 to_out"), (4, 16, 640, 640, True, "4 frames 16^2 to_out"),
    (1, 32, 320, 320, True, "1 frame 32^2 to_out"), (1, 32, 320, 2560, False, "1 frame 32^2 ff.net.0.proj"), (2, 16, 640, 640, True, "2 frames 16^2 to_out"),
    (16, 32, 512, 1536, False, "VAE mid attention to_qkv"), (16, 32, 512, 512, True, "VAE mid attention to_out"),
    (25, 8, 384, 25600, False, "audio context k | v, 1600 rows"),
    (16, 8, 1280, 1280, True, "8^2 to_out / to_q / proj"), (16, 8, 1280, 3840, False, "8^2 to_qkv"), (16, 8, 1280, 10240, False, "8^2 ff.net.0.proj"),
    (16, 32, 1280, 320, True, "32^2 ff.net.2"), (16, 16, 1280, 640, False, "16^2 shortcut 1280 -> 640"), (64, 8, 1280, 1280, True, "64 frames 8^2 to_out"),
    (8, 8, 1280, 1280, True, "8 frames 8^2 to_out"),
    (16, 16, 2560, 640, False, "16^2 ff.net.2 (lin_mp)"), (16, 8, 5120, 1280, False, "8^2 ff.net.2 (lin_mp)"), (16, 8, 2560, 1280, False, "8^2 shortcut 2560 -> 1280 (lin_mp)"),
    (64, 16, 2560, 640, False, "64 frames 16^2 ff.net.2 (lin_mp)"), (64, 8, 5120, 1280, False, "64 frames 8^2 ff.net.2 (lin_mp)"),
]


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    blocks = [v for v in sys.argv[2:]] or ["512"]          # LIN_FK_BLOCKS values
    eng = Engine(0)
    if os.environ.get("LIN_MP"):
        Engine.set_knob("LIN_MP", int(os.environ["LIN_MP"]))
    g = torch.Generator(device="cpu").manual_seed(1)
    settings = [("conv3", 0, "0")] + [(f"lin_fk/{b}", 1, b) for b in blocks]
    print("us per launch: " + " | ".join(n for n, _, _ in settings))
    for N, H, Cin, Cout, res, name in SHAPES:
        x = to_cb16(torch.randn(N, Cin, H, H, generator=g).cuda())
        w = (torch.randn(Cout, Cin, 1, 1, generator=g) * (2.0 / Cin) ** 0.5).numpy()
        y = empty_cb16(N, Cout, H, H)
        t = {n: [] for n, _, _ in settings}
        for rnd in range(3):
            for n, fk, b in settings:
                Engine.set_knob("LIN_FK", fk)
                if fk:
                    Engine.set_knob("LIN_FK_BLOCKS", int(b.split(":")[0]))
                ms = eng.conv2d_f16(x.data_ptr(), N, H, H, Cin, w, Cout, 1, 1, 0, False, 0, None, None, x.data_ptr() if res and Cin == Cout else 0, False,
                                    y.data_ptr(), iters)
                t[n].append(ms * 1e3)
        gf = 2.0 * N * H * H * Cin * Cout / 1e9
        print(f"{name:32s} M={N * H * H:6d} K={Cin} N={Cout:5d} {gf:6.1f} GFLOP  " + " | ".join(f"{np.median(t[n]):7.1f}" for n, _, _ in settings))
    eng.close()


if __name__ == "__main__":
    main()
