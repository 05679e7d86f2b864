#!/usr/bin/env python3
"""In-job A/B of whole Wav2Lip passes (ltk_wav2lip_time_convs: the device side of one ltk_wav2lip_infer pass, HIP events) under several
settings of the launch-time knobs, INTERLEAVED (A, B, C, A, B, C, ...): medians and minima per setting and frame count.  GPU only.

    ROUNDS=7 python scripts/pass_ab.py "DF_FRAMES=0" "DF_FRAMES=16,DF_MIN=32" "DF_FRAMES=32,DF_MIN=32" -- 32 64 256
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
    rounds = int(os.environ.get("ROUNDS", "7"))
    iters = int(os.environ.get("ITERS", "10"))
    eng = Engine(0)
    eng.load_wav2lip(synth.wav2lip_state_dict(1234), max_frames=max(frames))
    print("settings: " + " | ".join(args[:cut]))
    # every knob any setting touches is reset to the FIRST setting's value (or its value there = the default) before a setting is applied
    for nf in frames:
        t = [[] for _ in settings]
        for _ in range(rounds):
            for si, st in enumerate(settings):
                for k, v in settings[0]:
                    Engine.set_knob(k, v)
                for k, v in st:
                    Engine.set_knob(k, v)
                t[si].append(eng.time_convs(nf, iters)[0] * 1e3)
                for k, v in st:                      # knobs the first setting does not name go back to their defaults
                    if k not in dict(settings[0]):
                        Engine.set_knob(k, DEFAULTS[k])
        med = [float(np.median(x)) for x in t]
        mn = [float(np.min(x)) for x in t]
        print(f"{nf:4d} frames  median us: " + " ".join(f"{m:9.1f}" for m in med) + "   min us: " + " ".join(f"{m:9.1f}" for m in mn) +
              "   vs first: " + " ".join(f"{100 * (m / med[0] - 1):+5.1f}%" for m in med), flush=True)
    eng.close()


DEFAULTS = {"AUDIO_ROWCONV": 54, "PREFETCH": 1, "FACE_CACHE": 0, "DF_FRAMES": 0, "DF_MIN": 32, "DF_BLOCK": 6, "ROWCONV": 1024, "GRAPH": 1, "ROWGEMM": 1, "ROWCONVT": 512, "LDS_SWZ": 1, "CONV_S2SPLIT": 1}

if __name__ == "__main__":
    main()
