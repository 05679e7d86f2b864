#!/usr/bin/env python3
"""A few ltk_musetalk_infer calls of S sessions x 16 frames in one call (S x 16 frames per pass), for kernel traces of a controlled call size:
    rocprofv3 --kernel-trace ... -- python scripts/mt_call.py [sessions] [calls]        (environment knobs pass through; FP8=1)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import synth_inputs as synth  # noqa: E402
from livetalking_amd.engine import Engine  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 4
Bf, n = 16, 5
eng = Engine(0)
eng.load_musetalk(synth.musetalk_unet_state_dict(), synth.vae_decoder_state_dict(), max_frames=S * Bf, fp8=bool(int(os.environ.get("FP8", "0"))))
lats = synth.musetalk_latents(n)
frames, _, _ = synth.wav2lip_avatar(n_frames=n, full_hw=(360, 640), box=160, seed=2)
aid = eng.register_musetalk_avatar(lats, frames, [(240, 100, 400, 280)] * n, [np.full((255, 220, 3), 255, np.uint8)] * n, [(210, 60, 430, 315)] * n)
feats = [torch.from_numpy(synth.musetalk_whisper_feats(Bf, seed=30 + s)).cuda() for s in range(S)]
out = torch.zeros(S, Bf, 256, 256, 3, dtype=torch.uint8, device="cuda")
for c in range(calls):
    eng.musetalk_infer([(aid, (3 * s + c) % 7, Bf, feats[s].data_ptr(), out[s].data_ptr()) for s in range(S)])
torch.cuda.synchronize()
eng.close()
