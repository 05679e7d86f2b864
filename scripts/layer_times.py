#!/usr/bin/env python3
"""Per-layer times of the Wav2Lip conv stack inside a whole pass (ltk_wav2lip_time_layers) under several settings of the
launch-time knobs (csrc/tune.h), measured INTERLEAVED (setting A, B, C, A, B, C, ...) and reported as medians: clocks
drift by several per cent within a job, so back-to-back blocks of one setting are not comparable.  GPU only.

    python scripts/layer_times.py "RING=0" "RING=4,RING_KB=78" "RING=8,RING_KB=150" -- 16 256
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import synth_inputs as synth  # noqa: E402
from livetalking_amd.engine import Engine  # noqa: E402


def parse(spec):
    return [(kv.split("=")[0], int(kv.split("=")[1])) for kv in spec.split(",") if kv]


def main():
    args = sys.argv[1:]
    cut = args.index("--") if "--" in args else len(args)
    settings = [parse(a) for a in args[:cut]]
    frames = [int(x) for x in args[cut + 1:]] or [16]
    rounds = int(os.environ.get("ROUNDS", "5"))
    eng = Engine(0)
    eng.load_wav2lip(synth.wav2lip_state_dict(1234), max_frames=max(frames))
    names = eng.layer_names()
    base = settings[0]
    for nf in frames:
        per = [[] for _ in settings]
        stack = [[] for _ in settings]
        for _ in range(rounds):
            for si, st in enumerate(settings):
                for k, v in base:
                    Engine.set_knob(k, v)
                for k, v in st:
                    Engine.set_knob(k, v)
                eng.time_layers(nf, 1)
                per[si].append(eng.time_layers(nf, 4) * 1e3)
                stack[si].append(eng.time_convs(nf, 10)[0] * 1e3)
        med = [np.median(np.stack(p), axis=0) for p in per]
        print(f"==== {nf} frames, median us per layer over {rounds} interleaved rounds: " + " | ".join(args[:cut]))
        for i, n in enumerate(names):
            row = "".join(f"{m[i]:9.1f}" for m in med)
            d = [m[i] - med[0][i] for m in med[1:]]
            mark = "  <--" if any(abs(x) > max(1.0, 0.05 * med[0][i]) for x in d) else ""
            print(f"{n:28s}{row}{mark}")
        print(f"{'sum (one stream)':28s}" + "".join(f"{m.sum():9.1f}" for m in med))
        print(f"{'conv stack (two streams)':28s}" + "".join(f"{np.median(s):9.1f}" for s in stack), flush=True)
    for k, v in base:
        Engine.set_knob(k, v)
    eng.close()


if __name__ == "__main__":
    main()
