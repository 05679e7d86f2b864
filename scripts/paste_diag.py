#!/usr/bin/env python3
"""Diagnostic: ltk_paste_back against oracle/paste_oracle.py on TEXTURED predictions (the synthetic face crops / uniform noise) for a
spread of boxes; prints where and by how much the bytes differ.  GPU only."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import synth_inputs as synth  # noqa: E402
from livetalking_amd.engine import Engine  # noqa: E402
from oracle import paste_oracle  # noqa: E402


def main():
    eng = Engine(0)
    hw = (360, 640)
    frames, faces, coords = synth.wav2lip_avatar(n_frames=5, full_hw=hw, box=160, seed=0)
    rng = np.random.default_rng(1)
    boxes = [tuple(coords[0]), (10, 170, 20, 180), (0, 301, 640 - 333, 640), (50, 306, 100, 356), (100, 228, 200, 328), (5, 205, 7, 330), (3, 320, 1, 600)]
    preds = {"face": np.ascontiguousarray(faces[1]), "noise": rng.integers(0, 256, (256, 256, 3), dtype=np.uint8),
             "ramp": np.repeat(np.arange(256, dtype=np.uint8)[None, :, None], 256, 0).repeat(3, 2)}
    for bi, box in enumerate(boxes):
        cs = [box] * 5
        aid = eng.register_avatar(faces, frames, cs)
        for name, pred in preds.items():
            d_pred = torch.from_numpy(pred).cuda()
            out = np.empty((hw[0], hw[1], 3), dtype=np.uint8)
            eng.paste_back(aid, 0, d_pred.data_ptr(), out)
            ref = paste_oracle.paste_back_frame(pred.astype(np.float32), frames[0], box)
            d = np.abs(out.astype(np.int32) - ref.astype(np.int32))
            nz = np.argwhere(d > 0)
            msg = f"box {box} ({box[1]-box[0]}x{box[3]-box[2]}) pred {name}: differing bytes {len(nz)} max {d.max()}"
            if len(nz):
                y, x, c = nz[0]
                msg += f"; first at (y={y}, x={x}, c={c}) gpu {out[y, x, c]} ref {ref[y, x, c]}; rows {sorted(set(nz[:, 0]))[:8]} cols {sorted(set(nz[:, 1]))[:8]}"
            print("[paste-diag]", msg, flush=True)
        eng.release_avatar(aid)
    eng.close()


if __name__ == "__main__":
    main()
