#!/usr/bin/env python3
"""Host wall time of ltk_wav2lip_infer for the first (eager), second (hipGraph capture + instantiate + replay) and later (replay) call of a frame
count, per frame count.  GPU only."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import synth_inputs as synth  # noqa: E402
from livetalking_amd.engine import Engine  # noqa: E402


def main():
    eng = Engine(0)
    eng.load_wav2lip(synth.wav2lip_state_dict(1234), max_frames=256)
    frames, faces, coords = synth.wav2lip_avatar(n_frames=5, full_hw=(360, 640), box=160, seed=0)
    aid = eng.register_avatar(faces, frames, coords)
    mel = torch.randn(256, 80, 16, device="cuda")
    pred = torch.zeros(256, 256, 256, 3, dtype=torch.uint8, device="cuda")
    for nf in (16, 48, 128, 256):
        ts = []
        for it in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.wav2lip_infer([(aid, 0, nf, mel.data_ptr(), pred.data_ptr())])
            ts.append((time.perf_counter() - t0) * 1e3)
        print(f"[graph-cost] {nf:3d} frames: call 1 (eager) {ts[0]:.2f} ms, call 2 (capture + instantiate + replay) {ts[1]:.2f} ms, calls 3-5 (replay) "
              f"{ts[2]:.2f} {ts[3]:.2f} {ts[4]:.2f} ms; graphs {eng.graph_count()}", flush=True)
    eng.close()


if __name__ == "__main__":
    main()
