#!/usr/bin/env python3
"""conv3's 1x1 launches: the two-stage chunk loop (LTK_RING1 = 0) against the chunk ring (LTK_RING1 = 3, 4, 6 LDS stages), per layer shape:
outputs compared BIT FOR BIT (same chunk and MFMA order) and the launch timed (ltk_conv2d_f16, HIP events, 20 iterations).  GPU only.
The ring kernels live in commit cb1ad49 only (measured slower, profiles/r04_ring1_negative.txt): on a later tree the knob is unknown and this script fails.

    python scripts/ring1_ab.py [quick]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from livetalking_amd.engine import Engine  # noqa: E402
from livetalking_amd.layout import empty_cb16, to_cb16  # noqa: E402

# (N, H, W, Cin, Cout): MuseTalk's U-Net linears at 16 frames (tokens = N*H*W), small / ragged launches for the masks and the small-batch tiles
CASES = [(16, 32, 32, 320, 320), (16, 32, 32, 320, 2560), (16, 32, 32, 1280, 320), (16, 16, 16, 640, 640), (16, 16, 16, 2560, 640),
         (16, 8, 8, 1280, 1280), (16, 8, 8, 1280, 2560), (16, 8, 8, 5120, 1280), (2, 33, 17, 128, 80), (3, 5, 7, 64, 96), (3, 1, 1, 512, 512),
         (19, 1, 1, 512, 512), (4, 64, 64, 64, 128)]
DEPTHS = (0, 3, 4, 6)


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    eng = Engine(0)
    g = torch.Generator(device="cpu").manual_seed(7)
    print("N,H,W,Cin,Cout: us per launch at LTK_RING1 = " + " / ".join(str(d) for d in DEPTHS) + "   bitwise vs 0")
    worst = True
    for case in (CASES[:8:2] + CASES[8:] if quick else CASES):
        N, H, W, Cin, Cout = case
        x = torch.randn(N, Cin, H, W, generator=g).half().float()
        w = (torch.randn(Cout, Cin, 1, 1, generator=g) * (2.0 / Cin) ** 0.5).half().float().numpy()
        scale = (torch.rand(Cout, generator=g) + 0.5).numpy()
        shift = (torch.randn(Cout, generator=g) * 0.1).numpy()
        xd = to_cb16(x.cuda())
        outs, times = [], []
        for d in DEPTHS:
            Engine.set_knob("RING1", d)
            y = empty_cb16(N, Cout, H, W, fill=float("nan"))
            eng.conv2d_f16(xd.data_ptr(), N, H, W, Cin, w, Cout, 1, 1, 0, False, 0, scale, shift, 0, True, y.data_ptr())
            torch.cuda.synchronize()
            ms = eng.conv2d_f16(xd.data_ptr(), N, H, W, Cin, w, Cout, 1, 1, 0, False, 0, scale, shift, 0, True, y.data_ptr(), iters=20)
            torch.cuda.synchronize()
            outs.append(y.clone())
            times.append(ms * 1e3)
        same = [bool(torch.equal(outs[0].view(torch.int16), o.view(torch.int16))) for o in outs[1:]]
        finite = bool(torch.isfinite(outs[0].float()).all())
        worst = worst and all(same) and finite
        print(f"{case}: " + " / ".join(f"{t:7.1f}" for t in times) + f"   {same} finite={finite}", flush=True)
    Engine.set_knob("RING1", 0)
    print("ALL BITWISE EQUAL" if worst else "MISMATCH")
    eng.close()


if __name__ == "__main__":
    main()
