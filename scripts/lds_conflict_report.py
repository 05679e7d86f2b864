#!/usr/bin/env python3
"""Per-kernel LDS bank-conflict share (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, summed over the dispatches of a run) of two or more rocprofv3
--pmc output directories, side by side (one directory per knob setting of the same command).

    python scripts/lds_conflict_report.py LABEL=DIR [LABEL=DIR ...]
"""
import collections
import glob
import os
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    return name.replace("void ", "").replace("ltk::", "")


def counters(d):
    out = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for path in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
        db = sqlite3.connect(path)
        for k, c, v, cnt in db.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name"):
            out[short(k)][c] += v
            n[short(k)] = max(n[short(k)], cnt)
    return out, n


def main():
    runs = [a.split("=", 1) for a in sys.argv[1:]]
    data = [(lab, *counters(d)) for lab, d in runs]
    kernels = sorted({k for _, c, _ in data for k in c if c[k].get("SQ_LDS_IDX_ACTIVE", 0) > 0},
                     key=lambda k: -max(c[k].get("SQ_LDS_IDX_ACTIVE", 0) for _, c, _ in data))
    print("# LDS conflict cycles / LDS active cycles per kernel, all dispatches of the run; columns: " + " | ".join(lab for lab, _, _ in data))
    print(f"{'kernel':70s} {'dispatches':>10s} " + " ".join(f"{lab + ' conflict/active (active Mcycles)':>40s}" for lab, _, _ in data))
    tot = [[0.0, 0.0] for _ in data]
    for k in kernels:
        cells = []
        for i, (_, c, _) in enumerate(data):
            act, con = c[k].get("SQ_LDS_IDX_ACTIVE", 0.0), c[k].get("SQ_LDS_BANK_CONFLICT", 0.0)
            tot[i][0] += act; tot[i][1] += con
            cells.append(f"{(con / act if act else 0):34.3f} ({act / 1e6:7.1f})")
        print(f"{k[:70]:70s} {data[0][2][k]:10d} " + " ".join(cells))
    print(f"{'all kernels':70s} {'':10s} " + " ".join(f"{(t[1] / t[0] if t[0] else 0):34.3f} ({t[0] / 1e6:7.1f})" for t in tot))


if __name__ == "__main__":
    main()
