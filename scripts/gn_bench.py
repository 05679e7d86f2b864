#!/usr/bin/env python3
"""GroupNorm(+SiLU) kernels on the MuseTalk VAE / U-Net shapes of a 16-frame pass (ltk_groupnorm_f16): us per run and GB/s of the algorithmic
bytes (one read + one write of the tensor) for the two-pass kernels (impl 1), the block-per-(image, group) kernel (2) and the one-pass
cooperative kernel (3), interleaved rounds.  GPU only.

    python scripts/gn_bench.py [frames] [iters]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from livetalking_amd.engine import Engine  # noqa: E402
from livetalking_amd.layout import empty_cb16, to_cb16  # noqa: E402

SHAPES = [(128, 256), (256, 256), (256, 128), (512, 128), (512, 64), (512, 32), (320, 32), (640, 16), (1280, 8)]


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    eng = Engine(0)
    print(f"{frames} frames, {iters} back-to-back runs, median of 3 interleaved rounds: us (GB/s of 1 read + 1 write)")
    for C, H in SHAPES:
        x = to_cb16(torch.randn(frames, C, H, H).cuda())
        y = empty_cb16(frames, C, H, H)
        ga, be = np.ones(C, np.float32), np.zeros(C, np.float32)
        impls = [i for i in (1, 2, 3)]
        t = {i: [] for i in impls}
        for _ in range(3):
            for i in impls:
                try:
                    t[i].append(eng.groupnorm_f16(x.data_ptr(), frames, C, H * H, 32, 1e-6, ga, be, True, y.data_ptr(), impl=i, iters=iters) * 1e3)
                except RuntimeError:
                    t[i].append(float("nan"))
        gb = 2 * frames * C * H * H * 2 / 1e9
        cells = []
        for i in impls:
            m = float(np.median(t[i]))
            cells.append(f"impl {i}: {'   -   ' if m != m else f'{m:7.1f} us ({gb / (m * 1e-6):6.0f} GB/s)'}")
        print(f"{C:5d} ch @ {H:3d}^2  {gb * 1e3:7.1f} MB  " + "   ".join(cells))
    eng.close()


if __name__ == "__main__":
    main()
