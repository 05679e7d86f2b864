#!/usr/bin/env python3
"""Per-layer tile / split sweep of the conv3 kernels over the Wav2Lip layer geometries (GPU only).

For every layer and frame count: time the layer back to back (HIP events, 10 launches) under each combination of the
conv3 knobs (csrc/tune.h: CONV_PXW, CONV3_NBT, CONV3_NC8, KSPLIT) and print us / TFLOP/s per cell plus the best cell.
The result is the evidence behind the per-shape table in conv3_launch.

    SWEEP_FRAMES=16,64,256 python scripts/conv_sweep2.py [substring ...]
"""
import itertools
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from livetalking_amd.engine import Engine  # noqa: E402

# name, H, W, Cin, Cout, k, stride, pad, transposed, out_pad, residual
LAYERS = [
    ("c32@128", 128, 128, 32, 32, 3, 1, 1, False, 0, True),
    ("c64@64", 64, 64, 64, 64, 3, 1, 1, False, 0, True),
    ("c128@32", 32, 32, 128, 128, 3, 1, 1, False, 0, True),
    ("c256@16", 16, 16, 256, 256, 3, 1, 1, False, 0, True),
    ("c512@8", 8, 8, 512, 512, 3, 1, 1, False, 0, True),
    ("c512@4", 4, 4, 512, 512, 3, 1, 1, False, 0, True),
    ("c512@16", 16, 16, 512, 512, 3, 1, 1, False, 0, True),
    ("c384@32", 32, 32, 384, 384, 3, 1, 1, False, 0, True),
    ("c256@64", 64, 64, 256, 256, 3, 1, 1, False, 0, True),
    ("c128@128", 128, 128, 128, 128, 3, 1, 1, False, 0, True),
    ("c64@256", 256, 256, 64, 64, 3, 1, 1, False, 0, True),
    ("out80>32@256", 256, 256, 80, 32, 3, 1, 1, False, 0, False),
    ("T1024>512@4", 4, 4, 1024, 512, 3, 2, 1, True, 1, False),
    ("T1024>512@8", 8, 8, 1024, 512, 3, 2, 1, True, 1, False),
    ("T768>384@16", 16, 16, 768, 384, 3, 2, 1, True, 1, False),
    ("T512>256@32", 32, 32, 512, 256, 3, 2, 1, True, 1, False),
    ("T320>128@64", 64, 64, 320, 128, 3, 2, 1, True, 1, False),
    ("T160>64@128", 128, 128, 160, 64, 3, 2, 1, True, 1, False),
    ("s2 256>512@16", 16, 16, 256, 512, 3, 2, 1, False, 0, False),
    ("s2 512>512@8", 8, 8, 512, 512, 3, 2, 1, False, 0, False),
    ("1x1 512@1", 1, 1, 512, 512, 1, 1, 0, False, 0, False),
    ("1x1 8192>512@1", 1, 1, 8192, 512, 1, 1, 0, False, 0, False),     # the flattened 4x4 valid conv
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
    frames = [int(x) for x in os.environ.get("SWEEP_FRAMES", "16").split(",")]
    only = sys.argv[1:]
    eng = Engine(0)
    results = []
    for N in frames:
        print(f"\n==== frames = {N}   cell = us/TFLOPs   knobs: pxw nbt nc8 ksplit (0 = heuristic)")
        for l in LAYERS:
            name, H, W, Cin, Cout, k, s, p, tr, op, res = l
            if only and not any(o in name for o in only):
                continue
            x = (torch.randn(N, H, W, (Cin + 15) // 16 * 16, device="cuda") * 0.5).half()
            wshape = (Cin, Cout, k, k) if tr else (Cout, Cin, k, k)
            w = (np.random.default_rng(0).standard_normal(wshape) * 0.05).astype(np.float32)
            if tr and s == 2:
                Ho, Wo = H * 2, W * 2
            elif tr:
                Ho, Wo = k, k
            else:
                Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
            y = torch.empty(N, Ho, Wo, (Cout + 15) // 16 * 16, dtype=torch.float16, device="cuda")
            sc, sf = np.ones(Cout, np.float32), np.zeros(Cout, np.float32)
            is3 = (k == 3 and s == 1 and not tr)
            pxws = (0, 1, 2) if is3 else (0,)
            nbts = (0, 1) if (Cout >= 64 and not tr) else (0,)
            nc8s = (0, 2, 4) if (is3 and Cin % 32 == 0) else (0,)
            ksps = (0, 1, 2, 4, 8, 16)
            cells = {}
            for pxw, nbt, nc8, ks in itertools.product(pxws, nbts, nc8s, ksps):
                Engine.set_knob("CONV_PXW", pxw); Engine.set_knob("CONV3_NBT", nbt)
                Engine.set_knob("CONV3_NC8", nc8); Engine.set_knob("KSPLIT", ks)
                try:
                    ms = eng.conv2d_f16(x.data_ptr(), N, H, W, Cin, w, Cout, k, s, p, tr, op, sc, sf,
                                        x.data_ptr() if res else 0, True, y.data_ptr(), iters=10)
                    cells[(pxw, nbt, nc8, ks)] = ms * 1e3
                except Exception as ex:  # noqa: BLE001
                    cells[(pxw, nbt, nc8, ks)] = None
            base = cells.get((0, 0, 0, 0))
            ok = {kk: v for kk, v in cells.items() if v is not None}
            best = min(ok, key=ok.get)
            fl = 2.0 * macs(l) * N
            print(f"{name:18s} heuristic {base:7.1f} us ({fl / base / 1e6:5.0f} TF)   best {ok[best]:7.1f} us ({fl / ok[best] / 1e6:5.0f} TF) at pxw/nbt/nc8/ks={best}", flush=True)
            top = sorted(ok, key=ok.get)[:6]
            print("      " + "  ".join(f"{kk}:{ok[kk]:.1f}" for kk in top), flush=True)
            results.append({"frames": N, "layer": name, "heuristic_us": base, "best_us": ok[best], "best": best,
                            "cells": {str(kk): v for kk, v in cells.items()}})
    for kname in ("CONV_PXW", "CONV3_NBT", "CONV3_NC8", "KSPLIT"):
        Engine.set_knob(kname, 0)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "conv_sweep2.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as f:
        json.dump(results, f)
    eng.close()


if __name__ == "__main__":
    main()
