#!/usr/bin/env python3
"""Per-op times of the MuseTalk launch program inside a whole pass (ltk_musetalk_time_ops), aggregated by op type and by
block, with the top ops listed; optional interleaved A/B of one launch-time knob.  GPU only.

    python scripts/mt_op_times.py [frames] [KNOB=a,b]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import synth_inputs as synth  # noqa: E402
from livetalking_amd.engine import Engine  # noqa: E402

TYPES = ["conv/linear", "GroupNorm", "LayerNorm", "attention", "GEGLU", "add-pos", "v-transpose"]


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    ab = None
    if len(sys.argv) > 2:          # KNOB=a,b or KNOB1=a,b+KNOB2=c,d (setting i = the i-th value of every knob)
        ks = [kv.split("=") for kv in sys.argv[2].split("+")]
        ab = ("+".join(k for k, _ in ks), list(zip(*[[int(v) for v in vs.split(",")] for _, vs in ks])), [k for k, _ in ks])
    t0 = time.time()
    eng = Engine(0)
    eng.load_musetalk(synth.musetalk_unet_state_dict(), synth.vae_decoder_state_dict(), max_frames=frames, fp8=bool(int(os.environ.get("FP8", "0"))))
    print(f"loaded in {time.time() - t0:.0f} s")
    ops = eng.musetalk_ops()
    settings = [None] if ab is None else ab[1]
    cols = []
    for rnd in range(3):
        for si, v in enumerate(settings):
            if ab is not None:
                for kn, kv in zip(ab[2], v):
                    Engine.set_knob(kn, kv)
            eng.musetalk_time_ops(frames, 1)
            t = eng.musetalk_time_ops(frames, 3) * 1e3
            if rnd == 0:
                cols.append([t])
            else:
                cols[si].append(t)
    med = [np.median(np.stack(c), axis=0) for c in cols]
    for si, v in enumerate(settings):
        m = med[si]
        print(f"\n==== {frames} frames" + ("" if ab is None else f", {ab[0]}={v}") + f": pass {m.sum():.0f} us over {len(ops)} ops")
        for ti, tn in enumerate(TYPES):
            sel = [i for i, (_, t) in enumerate(ops) if t == ti]
            if sel:
                print(f"  {tn:12s} {len(sel):4d} ops {m[sel].sum():9.0f} us  {100 * m[sel].sum() / m.sum():5.1f} %")
        part = {}
        for i, (n, _) in enumerate(ops):
            key = ("vae." if n.startswith(("decoder", "post_quant")) else "unet.") + ".".join(n.replace("decoder.", "").split(".")[:2])
            part[key] = part.get(key, 0.0) + m[i]
        print("  by block: " + "  ".join(f"{k}={v:.0f}" for k, v in part.items()))
        top = np.argsort(-m)[:25]
        for i in top:
            print(f"    {m[i]:8.1f} us  {TYPES[ops[i][1]]:12s} {ops[i][0]}")
        if os.environ.get("ALL_OPS"):          # every op in program order (ALL_OPS=<substring filter or 1>)
            flt = os.environ["ALL_OPS"]
            print("  ---- all ops in program order")
            for i, (n, t) in enumerate(ops):
                if flt == "1" or any(f in n for f in flt.split(",")):
                    print(f"    {i:4d} {m[i]:8.1f} us  {TYPES[t]:12s} {n}")
    if ab is not None and len(med) == 2:
        d = med[1] - med[0]
        print(f"\n==== ops that moved > 5 % ({ab[0]}={ab[1][0]} -> {ab[1][1]}); total {med[0].sum():.0f} -> {med[1].sum():.0f} us")
        for i in np.argsort(d):
            if abs(d[i]) > max(2.0, 0.05 * med[0][i]):
                print(f"    {med[0][i]:8.1f} -> {med[1][i]:8.1f}  {ops[i][0]}")
    eng.close()


if __name__ == "__main__":
    main()
