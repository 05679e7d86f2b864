#!/usr/bin/env python3
"""Per-kernel register / LDS / scratch / occupancy table of libltk_hip.so's device code, from the compiler itself (no GPU needed).

Every source of livetalking_amd/csrc/Makefile is compiled device-only for gfx950 with the Makefile's own flags plus
`-Rpass-analysis=kernel-resource-usage`; the remarks are parsed into one line per kernel:

    file  VGPRs  AGPRs  SGPRs  scratch bytes/lane  spilled VGPRs / SGPRs  LDS bytes/block (static)  waves/SIMD  kernel

Dynamic LDS (conv3 / lin_fk / attention tiles: set per launch, see the pass timelines' lds column) is not part of the static figure.
`!!` = scratch memory / spilled VGPRs; `s` = SGPRs spilled into VGPR lanes (v_writelane / v_readlane, no memory traffic).

    python scripts/kernel_resources.py > profiles/rNN_kernel_resources.txt
"""
import concurrent.futures
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "livetalking_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CXXFILT = "c++filt"            # (binutils: does not know _Float16's DF16_; such names are shortened by pretty() below)
BASE = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# the per-file additions of the Makefile (kept in step by tests/test_abi_and_host.py::test_kernel_resources_script_follows_the_makefile)
EXTRA = {"misc_kernels.hip": ["-ffp-contract=off"], "egress_kernels.hip": ["-ffp-contract=off"],
         "nn_kernels.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}
FIELDS = [("VGPRs", "VGPRs"), ("AGPRs", "AGPRs"), ("TotalSGPRs", "SGPRs"), ("ScratchSize [bytes/lane]", "scratch"), ("VGPRs Spill", "vspill"),
          ("SGPRs Spill", "sspill"), ("LDS Size [bytes/block]", "lds"), ("Occupancy [waves/SIMD]", "occ")]


def makefile_sources():
    with open(os.path.join(CSRC, "Makefile")) as f:
        m = re.search(r"^SRCS\s*=\s*(.+)$", f.read(), re.M)
    return m.group(1).split()


def compile_one(src):
    with tempfile.TemporaryDirectory() as d:
        cmd = [HIPCC] + BASE + EXTRA.get(src, []) + ["--cuda-device-only", "-c", os.path.join(CSRC, src), "-o", os.path.join(d, "x.co"),
                                                     "-Rpass-analysis=kernel-resource-usage"]
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=d)
    if r.returncode != 0:
        raise RuntimeError(f"{src}: hipcc failed\n{r.stderr[-2000:]}")
    kernels, cur = [], None
    for line in r.stderr.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            kernels.append(cur)
            continue
        m = re.search(r"remark:\s+(.+?): (\S+) \[-Rpass-analysis", line)
        if m and cur is not None:
            for key, short in FIELDS:
                if m.group(1) == key:
                    cur[short] = int(m.group(2))
    return src, kernels


def pretty(mangled, demangled):
    """Kernel name without its parameter list; names the demangler gave up on: namespace::name + the raw template-argument encoding."""
    if demangled != mangled:
        n = re.sub(r"^void ", "", demangled)
        depth = 0
        for i, ch in enumerate(n):
            depth += ch == "<"
            depth -= ch == ">"
            if ch == "(" and depth == 0:
                return n[:i]
        return n
    m = re.match(r"_ZN(\d+)", mangled)
    if not m:
        return mangled
    parts, pos = [], 3
    while pos < len(mangled) and mangled[pos].isdigit():
        q = pos
        while mangled[q].isdigit():
            q += 1
        ln = int(mangled[pos:q])
        parts.append(mangled[q:q + ln])
        pos = q + ln
    targs = ""
    if pos < len(mangled) and mangled[pos] == "I":                  # template arguments: I ... E (integral literals L<type><value>E)
        e = mangled.index("EE", pos) if "EE" in mangled[pos:] else pos
        lits = re.findall(r"L([a-z])(n?\d+)E", mangled[pos:e + 1])
        targs = "<" + ", ".join(("true" if v == "1" else "false") if t == "b" else v.replace("n", "-") for t, v in lits) + ">"
    return "::".join(parts) + targs


def main():
    srcs = makefile_sources()
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        results = dict(ex.map(compile_one, srcs))
    names = [k["name"] for s in srcs for k in results[s]]
    dem = subprocess.run([CXXFILT], input="\n".join(names), capture_output=True, text=True).stdout.splitlines() if names else []
    dem = dict(zip(names, dem))
    print("# hipcc --offload-arch=gfx950 -O3 (+ the Makefile's per-file flags) --cuda-device-only -Rpass-analysis=kernel-resource-usage")
    print("# gfx950: 512 registers per lane and SIMD shared by VGPRs + AGPRs (occupancy = waves per SIMD the allocation allows; a kernel's")
    print("# __launch_bounds__ and its LDS tile may set a lower one at run time); lds = STATIC bytes per block (dynamic tiles are set per launch)")
    print("# !! = scratch memory / spilled VGPRs; s = SGPRs spilled into VGPR lanes (v_writelane / v_readlane: no memory traffic)")
    print(f"# {'file':<20}{'VGPR':>5}{'AGPR':>5}{'SGPR':>5}{'scratch':>8}{'vspill':>7}{'sspill':>7}{'lds':>7}{'occ':>4}  kernel")
    bad = sg = 0
    for s in srcs:
        for k in sorted(results[s], key=lambda k: dem.get(k["name"], k["name"])):
            flag = "!!" if k.get("scratch", 0) or k.get("vspill", 0) else "s " if k.get("sspill", 0) else "  "
            bad += flag == "!!"
            sg += flag == "s "
            n = pretty(k["name"], dem.get(k["name"], k["name"]))
            print(f"{flag}{s:<20}{k.get('VGPRs', -1):>5}{k.get('AGPRs', -1):>5}{k.get('SGPRs', -1):>5}{k.get('scratch', -1):>8}{k.get('vspill', -1):>7}"
                  f"{k.get('sspill', -1):>7}{k.get('lds', -1):>7}{k.get('occ', -1):>4}  {n}")
    print(f"# {sum(len(v) for v in results.values())} kernels; {bad} with scratch / spilled VGPRs (!!), {sg} with SGPRs spilled into VGPR lanes only (s)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
