#!/usr/bin/env python3
"""Per-op roofline table of the MuseTalk resnet 3x3 convs with fp16 and with e4m3 operands (knob: engine loaded with fp8=True), one job:
op, shape, GFLOP per launch, us and TFLOP/s in both modes, fraction of the dense MFMA peak of the instruction each mode uses (fp16: 2.5 PF;
fp8 on v_mfma_f32_32x32x16_fp8_fp8: 2.5 PF - the non-scaled fp8 MFMA runs at the fp16 rate - , on the MX-scaled 32x32x64: 5 PF), the
algorithmic HBM bytes of the launch (input + output + weights) and the time those bytes take at 6 TB/s.  GPU only.

    python scripts/mt_fp8_table.py [frames]
"""
import os
import re
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import synth_inputs as synth  # noqa: E402
from livetalking_amd.engine import Engine  # noqa: E402


def level_hw(name):
    if name.startswith("decoder."):
        m = re.match(r"decoder\.up_blocks\.(\d)\.", name)
        return 32 if not m else (32, 64, 128, 256)[int(m.group(1))]
    m = re.match(r"down_blocks\.(\d)\.", name)
    if m:
        return (32, 16, 8, 4)[int(m.group(1))]
    m = re.match(r"up_blocks\.(\d)\.", name)
    if m:
        return (4, 8, 16, 32)[int(m.group(1))]
    return 4          # mid_block


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    usd, vsd = synth.musetalk_unet_state_dict(), synth.vae_decoder_state_dict()
    t = {}
    ops = None
    for fp8 in (False, True, False, True):
        eng = Engine(0)
        eng.load_musetalk(usd, vsd, max_frames=frames, fp8=fp8)
        ops = eng.musetalk_ops()
        eng.musetalk_time_ops(frames, 1)
        t.setdefault(fp8, []).append(eng.musetalk_time_ops(frames, 3) * 1e3)
        eng.close()
    t16, t8 = np.median(np.stack(t[False]), axis=0), np.median(np.stack(t[True]), axis=0)
    print(f"# {frames} frames; us per launch, medians of two engines per mode (per-op events); MX = Cin >= 512 (csrc/musetalk.hip add_conv2)")
    print(f"{'op':58s} {'Cin':>5s} {'Cout':>5s} {'HxW':>5s} {'GFLOP':>8s} {'us16':>8s} {'TF16':>6s} {'fr16':>5s} {'us8':>8s} {'TF8':>6s} {'fr8':>5s} {'x':>5s} {'MB16':>7s} {'MB8':>7s} {'hbm16us':>8s} {'hbm8us':>7s}")
    rows = []
    for i, (name, ty) in enumerate(ops):
        if ty != 0 or not re.search(r"resnets\.\d\.conv[12]$", name):
            continue
        sd = vsd if name.startswith("decoder.") else usd
        w = sd[name + ".weight"]
        cout, cin = w.shape[0], w.shape[1]
        if cin % 32:
            continue
        hw = level_hw(name)
        gflop = 2.0 * frames * hw * hw * cin * cout * 9 / 1e9
        mx = cin >= 512 and cin % 64 == 0
        peak8 = 5000.0 if mx else 2500.0
        px = frames * hw * hw
        mb16 = (px * cin * 2 + px * cout * 2 * (2 if name.endswith("conv2") else 1) + cin * cout * 9 * 2) / 1e6      # conv2 also reads the residual
        mb8 = (px * cin * 1 + px * cout * 2 * (2 if name.endswith("conv2") else 1) + cin * cout * 9 * 1) / 1e6
        rows.append((name, cin, cout, hw, gflop, t16[i], t8[i], peak8, mb16, mb8, mx))
    for name, cin, cout, hw, gflop, a, b, peak8, mb16, mb8, mx in rows:
        print(f"{name:58s} {cin:5d} {cout:5d} {hw:5d} {gflop:8.1f} {a:8.1f} {gflop / a * 1e3:6.0f} {gflop / a * 1e3 / 2500:5.2f} {b:8.1f} {gflop / b * 1e3:6.0f} "
              f"{gflop / b * 1e3 / peak8:5.2f} {a / b:5.2f} {mb16:7.1f} {mb8:7.1f} {mb16 / 6.0:8.1f} {mb8 / 6.0:7.1f}{'  MX' if mx else ''}")
    a = sum(r[5] for r in rows); b = sum(r[6] for r in rows)
    print(f"# fp8-capable convs: {a:.0f} -> {b:.0f} us ({a / b:.2f}x); whole pass {t16.sum():.0f} -> {t8.sum():.0f} us ({t16.sum() / t8.sum():.2f}x); GroupNorm ops "
          f"{sum(t16[i] for i, (_, ty) in enumerate(ops) if ty == 1):.0f} -> {sum(t8[i] for i, (_, ty) in enumerate(ops) if ty == 1):.0f} us")


if __name__ == "__main__":
    main()
