#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd (.db) outputs: per-kernel stats, per-layer conv durations of
one inference pass, and per-kernel PMC sums.  Usage: prof_report.py <dir-with-r_results.db> [--layers] [--csv out]"""
import argparse
import collections
import re
import sqlite3

LAYERS = ([f"audio_encoder.{i}" for i in range(13)]
          + ["face_encoder_blocks.%d.%d" % (b, j) for b, n in enumerate([1, 3, 4, 3, 3, 2, 2, 2]) for j in range(n)]
          + ["face_decoder_blocks.%d.%d" % (b, j) for b, n in enumerate([1, 2, 2, 3, 3, 3, 3, 3]) for j in range(n)]
          + ["output_block.0"])
MMAC = [0.4, 11.8, 11.8, 8.0, 15.9, 15.9, 4.0, 8.0, 8.0, 2.7, 5.3, 1.2, 0.3,
        308.3, 75.5, 151.0, 151.0, 75.5, 151.0, 151.0, 151.0, 75.5, 151.0, 151.0, 75.5, 151.0, 151.0, 75.5, 151.0, 37.7, 37.7, 4.2, 0.3,
        0.3, 8.4, 37.7, 75.5, 151.0, 302.0, 604.0, 604.0, 679.5, 1359.0, 1359.0, 1208.0, 2415.9, 2415.9, 1509.9, 2415.9, 2415.9,
        1509.9, 2415.9, 2415.9, 1509.9]


def short(name):
    name = re.sub(r"\(.*$", "", name)
    return name.replace("void ", "").replace("ltk::", "")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--layers", action="store_true")
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--csv")
    a = ap.parse_args()
    db = sqlite3.connect(f"{a.dir}/r_results.db")
    rows = db.execute("select name, start, end, grid_x, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, sgpr_count "
                      "from kernels order by start").fetchall()
    stats = collections.OrderedDict()
    for r in rows:
        k = short(r[0])
        s = stats.setdefault(k, [0, 0.0, 1e30, 0.0, r[6], r[7], r[8]])
        d = (r[2] - r[1]) / 1e3
        s[0] += 1; s[1] += d; s[2] = min(s[2], d); s[3] = max(s[3], d)
    tot = sum(s[1] for s in stats.values())
    out = ["kernel,calls,total_us,avg_us,min_us,max_us,pct,vgpr,agpr,sgpr"]
    for k, s in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        out.append(f"{k},{s[0]},{s[1]:.1f},{s[1]/s[0]:.2f},{s[2]:.2f},{s[3]:.2f},{100*s[1]/tot:.2f},{s[4]},{s[5]},{s[6]}")
    print("\n".join(out))
    if a.csv:
        open(a.csv, "w").write("\n".join(out) + "\n")
    if a.layers:
        idx = [i for i, r in enumerate(rows) if "pack_faces" in r[0]]
        if idx:
            i0 = idx[-1]
            seq = [r for r in rows[i0:] if "conv_mfma_kernel" in r[0]][:54]
            print(f"\nper-layer (last inference pass, {a.frames} frames): layer, us, TFLOP/s, grid, lds, kernel")
            tt = 0
            for name, mm, r in zip(LAYERS, MMAC, seq):
                d = (r[2] - r[1]) / 1e3
                tt += d
                print(f"{name:28s} {d:8.1f} {2*mm*1e6*a.frames/d/1e6:8.1f} {r[3]//r[4]:7d} {r[5]:7d}  {short(r[0])}")
            span = (seq[-1][2] - seq[0][1]) / 1e3
            print(f"sum of conv kernels {tt:.1f} us; first-start..last-end {span:.1f} us")
    pm = db.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
    if pm:
        print("\nPMC sums: kernel, counter, sum, dispatches")
        for r in pm:
            print(f"{short(r[0])},{r[1]},{r[2]:.0f},{r[3]}")


if __name__ == "__main__":
    main()
