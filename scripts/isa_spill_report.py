#!/usr/bin/env python3
"""Where a kernel's spill code sits: inside or outside its MFMA loops (gfx950 assembly of one source, no GPU needed).

scripts/kernel_resources.py lists which kernels spill SGPRs into VGPR lanes (v_writelane / v_readlane) or VGPRs into scratch; what that
costs depends on WHERE the spill code runs.  For every kernel of the given sources this prints its innermost loop that contains MFMAs
(smallest backward-branch range with a v_mfma in it) - instructions, MFMAs - and the spill instructions inside that loop, inside the
next enclosing MFMA loop (one work item of the persistent conv3 kernels) and in the whole kernel; and what the innermost loop waits on:
barriers, waits for ALL outstanding vector-memory loads (vmcnt(0)), LDS reads and LDS-DMA loads.

    python scripts/isa_spill_report.py conv3_mfma.hip conv_mfma.hip >> profiles/rNN_kernel_resources.txt
"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_resources import BASE, CSRC, EXTRA, HIPCC, pretty  # noqa: E402

SPILL = re.compile(r"v_readlane_b32|v_writelane_b32|scratch_load|scratch_store")
INSTR = re.compile(r"^\s+(v_|s_|ds_|global_|buffer_|scratch_|flat_)")


def asm_of(src):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "x.s")
        r = subprocess.run([HIPCC] + BASE + EXTRA.get(src, []) + ["--cuda-device-only", "-S", os.path.join(CSRC, src), "-o", out],
                           capture_output=True, text=True, cwd=d)
        if r.returncode != 0:
            raise RuntimeError(f"{src}: hipcc failed\n{r.stderr[-2000:]}")
        with open(out) as f:
            return f.read().split("\n")


def kernels(lines):
    cur, start = None, 0
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            cur, start = m.group(1), i
        elif cur and l.strip().startswith(".Lfunc_end"):
            yield cur, lines[start:i]
            cur = None


def report(name, body):
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    loops = []
    for i, l in enumerate(body):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and labels.get(m.group(1), i) < i:
            loops.append((labels[m.group(1)], i))
    def count(x, y):
        seg = body[x:y + 1]
        return (sum(bool(INSTR.match(l)) for l in seg), sum("v_mfma" in l for l in seg), sum(bool(SPILL.search(l)) for l in seg))

    def sync(x, y):          # what makes a wave wait inside the loop: barriers, waits for ALL outstanding vector-memory loads, LDS reads / DMA issues
        seg = body[x:y + 1]
        return (sum("s_barrier" in l for l in seg), sum(bool(re.search(r"s_waitcnt\s+.*vmcnt\(0\)", l)) for l in seg),
                sum(bool(re.search(r"\sds_read|\sds_load", l)) for l in seg), sum(bool(re.search(r"global_load_lds|buffer_load.* lds", l)) for l in seg))
    mf = sorted(((y - x, x, y) for x, y in loops if any("v_mfma" in l for l in body[x:y + 1])))
    total = sum(bool(SPILL.search(l)) for l in body)
    if not mf:
        nm = sum("v_mfma" in l for l in body)
        return f"{pretty(name, name):<46} no MFMA loop ({nm} MFMAs in straight-line code); spill instructions in the kernel: {total}"
    _, x, y = mf[0]
    ins, nm, sp = count(x, y)
    outer = next(((a, b) for _, a, b in mf[1:] if a <= x and b >= y and count(a, b)[1] >= nm and (b - a) > 2 * (y - x)), None)
    o = count(*outer) if outer else None
    nb, nw, nds, ndma = sync(x, y)
    return (f"{pretty(name, name):<46} innermost MFMA loop: {ins:>4} instructions, {nm:>3} MFMAs, {sp:>2} spill instructions, "
            f"{nb} s_barrier, {nw} vmcnt(0) waits, {nds:>3} LDS reads, {ndma:>2} LDS-DMA loads | "
            + (f"enclosing loop: {o[0]:>5} instructions, {o[2]:>3} spill instructions | " if o else "") + f"whole kernel: {total}")


def main():
    srcs = sys.argv[1:] or ["conv3_mfma.hip", "conv_mfma.hip"]
    print("#")
    print("# scripts/isa_spill_report.py " + " ".join(srcs) + ": spill instructions (v_readlane / v_writelane / scratch_load / scratch_store) by loop level")
    print("# (lane moves a kernel issues on purpose - cross-lane reductions - count too: kernel_resources.py's spill columns say which kernels spill at all)")
    for s in srcs:
        lines = asm_of(s)
        names = [n for n, _ in kernels(lines)]
        dem = dict(zip(names, subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()))
        rows = []
        for n, body in kernels(lines):
            if not any(".amdhsa_kernel " + n in l for l in lines):      # device functions are not kernels
                continue
            rows.append(report(n, body).replace(pretty(n, n), pretty(n, dem.get(n, n)), 1))
        for r in sorted(rows):
            print(f"{s:<18}{r}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
