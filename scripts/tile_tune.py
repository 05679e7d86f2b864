#!/usr/bin/env python3
"""In-situ tile / split tuning of the Wav2Lip conv stack (GPU only).

For every conv3 layer and every frame-count bucket: set that ONE layer's entry of the engine's tile table to each
candidate (ltk_wav2lip_set_layer_tile), time the WHOLE pass layer by layer (ltk_wav2lip_time_layers: HIP events between
consecutive launches, so the layer sees the cache state its predecessor left) and keep the candidate with the
smallest time for that layer.  Tiles never change an output element's summation order; split factors do, so the
split is only tuned where the rule would split anyway.  Prints a C table for engine.hip and writes
gpurun_out/tile_tune.json.

    TUNE_FRAMES=16,32,64,128,256 python scripts/tile_tune.py
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import synth_inputs as synth  # noqa: E402
from livetalking_amd.engine import Engine  # noqa: E402

BUCKET = {16: 0, 32: 1, 64: 2, 128: 3, 256: 4}


def main():
    frames = [int(x) for x in os.environ.get("TUNE_FRAMES", "16,256").split(",")]
    iters = int(os.environ.get("TUNE_ITERS", "6"))
    eng = Engine(0)
    eng.load_wav2lip(synth.wav2lip_state_dict(1234), max_frames=max(frames))
    names = eng.layer_names()
    L = len(names)
    result = {}
    for nf in frames:
        b = BUCKET[nf]
        for i in range(L):
            eng.set_layer_tile(i, b, 0, 0, 0)
        base = eng.time_layers(nf, iters)
        base2 = eng.time_layers(nf, iters)
        noise = float(np.abs(base - base2).max())
        print(f"\n==== {nf} frames: rule pass {base.sum()*1e3:.0f} us (repeat {base2.sum()*1e3:.0f} us, per-layer noise <= {noise*1e3:.1f} us)")
        best = {}
        for i, name in enumerate(names):
            cands = [(0, 0, 0)] + [(p, n, 0) for p in (4, 2, 1) for n in (2, 1)]
            cands += [(0, 0, k) for k in (1, 2, 4, 8)]
            times = {}
            for c in cands:
                try:
                    eng.set_layer_tile(i, b, *c)
                    t = eng.time_layers(nf, iters)
                    times[c] = float(t[i])
                except Exception:  # noqa: BLE001 - candidate not available for this layer
                    pass
            eng.set_layer_tile(i, b, 0, 0, 0)
            t0 = times.get((0, 0, 0), float(base[i]))
            # tiles first (bit-identical results); a split only if it wins clearly
            tile_c = min((c for c in times if c[2] == 0), key=lambda c: times[c])
            pick = tile_c if times[tile_c] < t0 - max(0.0004, 0.03 * t0) else (0, 0, 0)
            ks_c = min((c for c in times if c[2] != 0), key=lambda c: times[c], default=None)
            if ks_c is not None and times[ks_c] < min(times[pick], t0) - max(0.0008, 0.06 * t0):
                pick = ks_c
            best[name] = {"pick": pick, "rule_us": t0 * 1e3, "pick_us": times.get(pick, t0) * 1e3,
                          "all": {str(c): round(v * 1e3, 2) for c, v in sorted(times.items(), key=lambda kv: kv[1])}}
            flag = "" if pick == (0, 0, 0) else f"  -> {pick} {times[pick]*1e3:.1f}"
            print(f"{name:28s} rule {t0*1e3:6.1f} us{flag}", flush=True)
        # apply all picks together and re-time
        for i, name in enumerate(names):
            eng.set_layer_tile(i, b, *best[name]["pick"])
        tuned = eng.time_layers(nf, iters)
        ms, _ = eng.time_convs(nf, 10)
        print(f"==== {nf} frames: tuned pass {tuned.sum()*1e3:.0f} us (rule {base.sum()*1e3:.0f}); two-stream conv stack {ms*1e3:.0f} us")
        result[nf] = {"rule_pass_us": base.sum() * 1e3, "tuned_pass_us": tuned.sum() * 1e3, "conv_stack_us": ms * 1e3, "layers": best}
    print("\n// ---- table for engine.hip (layer prefix, bucket, pxw, nbt, ksplit)")
    for nf in frames:
        for name, r in result[nf]["layers"].items():
            if tuple(r["pick"]) != (0, 0, 0):
                p = r["pick"]
                print(f'    {{"{name}", {BUCKET[nf]}, {p[0]}, {p[1]}, {p[2]}}},')
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "tile_tune.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as f:
        json.dump({str(k): v for k, v in result.items()}, f, indent=1)
    eng.close()


if __name__ == "__main__":
    main()
