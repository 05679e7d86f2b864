#!/usr/bin/env python3
"""What bounds conv3_kernel<1,2,4,2,9> - clock, or idle MFMA cycles?  (round-3 verdict, item 2)

Runs the REAL kernel (through ltk_conv2d_f16) on one layer with one operand fill per process, `iters` back-to-back launches:

    python scripts/conv_clock.py <random|constant|zeros> [layer] [iters]

layers: c256@64 (256 ch @ 64^2 x 16 frames: 16 channel chunks per item, MFMA-bound), c64@256 (64 ch @ 256^2: 4 chunks per item,
134 MB in + 134 MB out per launch), c128@128.  `scripts/gpu_job.sh clock` runs every (layer, fill) once un-profiled (HIP-event wall
time) and once under `rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_...`; scripts/clock_report.py turns the result databases
into profiles/r04_conv3_clock.txt: per fill wall time, effective clock (GRBM_GUI_ACTIVE / wall), MFMA-pipe busy share, and the
wave-cycle split (parked at s_waitcnt / barrier, issue-stalled, issuing).
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from livetalking_amd.engine import Engine  # noqa: E402

LAYERS = {"c256@64": (64, 256), "c64@256": (256, 64), "c128@128": (128, 128)}


def main():
    fill = sys.argv[1] if len(sys.argv) > 1 else "random"
    layer = sys.argv[2] if len(sys.argv) > 2 else "c256@64"
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    N = int(os.environ.get("CLOCK_FRAMES", "16"))
    HW, C = LAYERS[layer]
    rng = np.random.default_rng(0)
    if fill == "random":
        x = (torch.randn(N, C // 16, HW, HW, 16, device="cuda") * 0.5).half()
        w = (rng.standard_normal((C, C, 3, 3)) * 0.05).astype(np.float32)
    elif fill == "constant":
        x = torch.full((N, C // 16, HW, HW, 16), 0.5, device="cuda").half()
        w = np.full((C, C, 3, 3), 0.0117, np.float32)
    else:
        x = torch.zeros(N, C // 16, HW, HW, 16, device="cuda").half()
        w = np.zeros((C, C, 3, 3), np.float32)
    y = torch.empty_like(x)
    eng = Engine(0)
    sc, sf = np.ones(C, np.float32), np.zeros(C, np.float32)
    # warm the clocks / caches, then the measured run (every launch reads x, writes y: same traffic for every fill)
    eng.conv2d_f16(x.data_ptr(), N, HW, HW, C, w, C, 3, 1, 1, False, 0, sc, sf, 0, True, y.data_ptr(), iters=50)
    ms = eng.conv2d_f16(x.data_ptr(), N, HW, HW, C, w, C, 3, 1, 1, False, 0, sc, sf, 0, True, y.data_ptr(), iters=iters)
    gflop = 2.0 * N * HW * HW * C * C * 9 / 1e9
    print(f"[clock] layer={layer} fill={fill} frames={N} iters={iters} us_per_launch={ms * 1e3:.2f} tflops={gflop / ms:.1f} "
          f"frac_of_2.5PF={gflop / ms / 2500:.3f}", flush=True)
    eng.close()


if __name__ == "__main__":
    main()
