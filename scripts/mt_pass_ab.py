#!/usr/bin/env python3
"""In-job A/B of whole MuseTalk passes (ltk_musetalk_time: U-Net + VAE decoder through run_program, i.e. the captured hipGraph,
HIP events) under several knob settings, INTERLEAVED (A, B, C, A, B, C, ...): medians and minima per setting and frame count.
A setting is a comma-separated list KNOB=value; knobs that are read when the program is BUILT (MT_FUSE) get an engine of
their own per distinct value, launch-time knobs are switched in place.  `UNET=1` in a setting times the U-Net ops only (per-op
events: no graph, no side branch).  GPU only.

    ROUNDS=5 python scripts/mt_pass_ab.py "MT_FUSE=0" "MT_FUSE=1" "MT_FUSE=3" "MT_FUSE=3,MT_GN1=0" -- 16 64
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import synth_inputs as synth  # noqa: E402
from livetalking_amd.engine import Engine  # noqa: E402

BUILD_KNOBS = ("MT_FUSE", "MT_ROWCONV", "GEMM_NC8")
DEFAULTS = {"MT_GN1": 1, "GRAPH": 1, "MT_TILE_TABLE": 1, "LIN_FK": 1, "LIN_FK_BLOCKS": 512, "LIN_FK_MIN_ROWS": 512, "ATTN_LDS": 1, "LIN_MP": 1, "GN_COOP": 1}


def parse(spec):
    return [(kv.split("=")[0], int(kv.split("=")[1])) for kv in spec.split(",") if kv]


def main():
    args = sys.argv[1:]
    cut = args.index("--") if "--" in args else len(args)
    settings = [parse(a) for a in args[:cut]] or [[]]
    frames = [int(x) for x in args[cut + 1:]] or [16]
    rounds = int(os.environ.get("ROUNDS", "5"))
    iters = int(os.environ.get("ITERS", "5"))
    fp8 = bool(int(os.environ.get("FP8", "0")))
    usd, vsd = synth.musetalk_unet_state_dict(), synth.vae_decoder_state_dict()
    engines = {}

    def engine_for(st):
        key = tuple((k, v) for k, v in st if k in BUILD_KNOBS)
        if key not in engines:
            for k, v in key:
                Engine.set_knob(k, v)
            e = Engine(0)
            e.load_musetalk(usd, vsd, max_frames=max(frames), fp8=fp8)
            engines[key] = e
        return engines[key]

    print("settings: " + " | ".join(args[:cut]), flush=True)
    for nf in frames:
        t = [[] for _ in settings]
        for _ in range(rounds):
            for si, st in enumerate(settings):
                eng = engine_for(st)
                for k, v in st:
                    if k not in BUILD_KNOBS and k != "UNET":
                        Engine.set_knob(k, v)
                if dict(st).get("UNET"):
                    ops = eng.musetalk_ops()
                    eng.musetalk_time_ops(nf, 1)
                    per = eng.musetalk_time_ops(nf, iters) * 1e3
                    t[si].append(float(sum(per[i] for i, (n, _) in enumerate(ops) if not n.startswith(("decoder.", "post_quant")))))
                else:
                    eng.musetalk_time(nf, 2)
                    t[si].append(eng.musetalk_time(nf, iters)[0] * 1e3)
                for k, v in st:
                    if k not in BUILD_KNOBS and k != "UNET" and k in DEFAULTS:
                        Engine.set_knob(k, DEFAULTS[k])
        med = [float(np.median(x)) for x in t]
        mn = [float(np.min(x)) for x in t]
        print(f"{nf:4d} frames  median us: " + " ".join(f"{m:9.1f}" for m in med) + "   min us: " + " ".join(f"{m:9.1f}" for m in mn) +
              "   vs first: " + " ".join(f"{100 * (m / med[0] - 1):+5.1f}%" for m in med), flush=True)
    for e in engines.values():
        e.close()


if __name__ == "__main__":
    main()
