#!/usr/bin/env python3
"""Probe: does a 16-frame Wav2Lip pass finish sooner as two concurrent 8-frame passes (two engines = two stream sets on one
GPU, one thread each) than as one 16-frame pass?  The small-map half of a pass is a chain of dependent launches that leave
most CUs idle; a second chain can fill them.  GPU only.

    python scripts/two_stream_probe.py [iters]
"""
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import synth_inputs as synth  # noqa: E402
from livetalking_amd.engine import Engine  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    frames, faces, coords = synth.wav2lip_avatar(n_frames=32, full_hw=(720, 1280), box=320, seed=0)
    weights = synth.wav2lip_state_dict(1234)
    engs = []
    for _ in range(2):
        e = Engine(0)
        e.load_wav2lip(weights, max_frames=32)
        engs.append((e, e.register_avatar(faces, frames, coords)))
    mel = torch.rand(32, 80, 16, device="cuda") * 4 - 2
    pred = torch.empty(2, 32, 256, 256, 3, dtype=torch.uint8, device="cuda")

    def run(ei, nf, n):
        e, aid = engs[ei]
        for i in range(n):
            e.wav2lip_infer([(aid, (i * 7) % 32, nf, mel[ei * 16:].data_ptr(), pred[ei].data_ptr())])

    def timed(jobs, n):
        for ei, nf in jobs:
            run(ei, nf, 5)
        torch.cuda.synchronize()
        ths = [threading.Thread(target=run, args=(ei, nf, n)) for ei, nf in jobs]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return sum(nf for _, nf in jobs) * n / dt, dt / n * 1e3

    for rnd in range(3):
        for name, jobs in (("one engine, 16 frames per call", [(0, 16)]),
                           ("two engines, 8 frames per call each, concurrently", [(0, 8), (1, 8)]),
                           ("one engine, 8 frames per call", [(0, 8)]),
                           ("one engine, 32 frames per call", [(0, 32)]),
                           ("two engines, 16 frames per call each, concurrently", [(0, 16), (1, 16)])):
            fps, ms = timed(jobs, iters)
            print(f"round {rnd}: {name:52s} {fps:9.0f} fps   {ms:7.3f} ms per round of calls", flush=True)
    for e, _ in engs:
        e.close()


if __name__ == "__main__":
    main()
