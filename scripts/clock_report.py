#!/usr/bin/env python3
"""gpurun_out/clock/<layer>_<fill>/r_results.db (rocprofv3 --kernel-trace --pmc ...) + the un-profiled log -> the table of
profiles/r04_conv3_clock.txt.  Units: GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed
over the 1024 SIMDs; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_ANY count quad-cycles summed over waves (MI355X_MICROARCH.md)."""
import os
import re
import sqlite3
import sys


def main():
    src = sys.argv[1]
    log = open(os.path.join(src, "unprofiled.log")).read() if os.path.exists(os.path.join(src, "unprofiled.log")) else ""
    wall = {}
    for m in re.finditer(r"layer=(\S+) fill=(\S+) frames=(\d+) iters=(\d+) us_per_launch=([\d.]+) tflops=([\d.]+)", log):
        wall[(m.group(1), m.group(2))] = (float(m.group(5)), float(m.group(6)))
    print("# conv3_kernel<1,2,4,2,9,1,0> (the real kernel, ltk_conv2d_f16, 16 frames, 200 back-to-back launches per run), one run per operand fill.")
    print("# un-profiled: HIP events around the 200 launches.  profiled: rocprofv3 --kernel-trace --pmc (kernel durations and counters of the SAME run;")
    print("#   the profiler serialises dispatches, so its per-kernel time has no launch overlap).  clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel time;")
    print("#   mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8);  wave split = share of SQ_WAVE_CYCLES.")
    print(f"{'layer':10s} {'fill':9s} {'us(unprof)':>10s} {'TF/s':>7s} {'us(prof)':>9s} {'clock GHz':>9s} {'mfma_busy':>9s} {'mfma cyc/launch/SIMD':>21s} "
          f"{'WAIT_ANY':>9s} {'WAIT_INST':>9s} {'ACTIVE':>7s}")
    for d in sorted(os.listdir(src)):
        db_path = os.path.join(src, d, "r_results.db")
        if not os.path.exists(db_path):
            continue
        layer, fill = d.rsplit("_", 1)
        db = sqlite3.connect(db_path)
        rows = db.execute("select name, start, end from kernels where name like '%conv3_kernel%' order by start").fetchall()
        rows = rows[len(rows) // 4:]                         # skip the warm-up launches
        us = sum(r[2] - r[1] for r in rows) / 1e3 / max(len(rows), 1)
        c = {}
        for name, val, n in db.execute("select counter_name, sum(value), count(*) from counters_collection where kernel_name like '%conv3_kernel%' "
                                       "group by counter_name"):
            c[name] = (val, n)
        nd = max(c.get("GRBM_GUI_ACTIVE", (0, 1))[1], 1)
        gui = c.get("GRBM_GUI_ACTIVE", (0, 1))[0] / nd / 8.0           # cycles per dispatch per XCD
        mfma = c.get("SQ_VALU_MFMA_BUSY_CYCLES", (0, 1))[0] / nd / 1024.0
        wave = c.get("SQ_WAVE_CYCLES", (0, 1))[0]
        sp = [c.get(k, (0, 1))[0] / wave if wave else float("nan") for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")]
        # per-dispatch durations of the pmc run (all dispatches, warm-up included, like the counter sums)
        allrows = db.execute("select start, end from kernels where name like '%conv3_kernel%'").fetchall()
        us_all = sum(e - s for s, e in allrows) / 1e3 / max(len(allrows), 1)
        w = wall.get((layer, fill), (float("nan"), float("nan")))
        print(f"{layer:10s} {fill:9s} {w[0]:10.2f} {w[1]:7.1f} {us:9.2f} {gui / (us_all * 1e3):9.3f} {mfma / gui if gui else float('nan'):9.3f} {mfma:21.0f} "
              f"{sp[0]:9.3f} {sp[1]:9.3f} {sp[2]:7.3f}")


if __name__ == "__main__":
    main()
