"""Regression guard for the LDS-DMA pipelines (global_load_lds): a wave must wait for its own tile copies (`s_waitcnt vmcnt`) before
the workgroup barrier that publishes the tile.  hipcc derives that wait from __syncthreads() for conv3's chunk loop, but hoisted it
OUT of attn_lds_kernel's tile loop (round 6: a second call with equal inputs gave other frames; profiles/r06_attn_lds_ab.txt).  The
kernels that stage through a DMA ring now carry the wait as inline asm; this test compiles the two sources to gfx950 ISA (no GPU
needed) and checks every barrier of those kernels - and the chunk-loop barrier of every conv3 instantiation - for it."""
import os
import re
import shutil
import subprocess

import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "livetalking_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _functions(src, tmp_path):
    out = tmp_path / (os.path.basename(src) + ".s")
    extra = ["-mllvm", "-amdgpu-mfma-vgpr-form=1"] if src == "nn_kernels.hip" else []        # as csrc/Makefile builds it
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", *extra, "-S", "--cuda-device-only", "-I", CSRC, os.path.join(CSRC, src), "-o", str(out)],
                   check=True, capture_output=True, timeout=900)
    funcs, cur = {}, None
    for line in out.read_text().split("\n"):
        m = re.match(r"^(_ZN3ltk\w+):", line)
        if m:
            cur = m.group(1)
            funcs[cur] = []
        if cur is not None:
            funcs[cur].append(line)
        if "s_endpgm" in line:
            cur = None
    return funcs


def _barrier_waits(lines):
    """per s_barrier: True when an s_waitcnt with a vmcnt field sits within the six instructions in front of it"""
    res = []
    for i, l in enumerate(lines):
        if "s_barrier" in l:
            res.append(any("s_waitcnt" in x and "vmcnt" in x for x in lines[max(0, i - 8):i]))
    return res


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_dma_ring_kernels_wait_for_their_copies_before_the_barrier(tmp_path):
    conv3 = _functions("conv3_mfma.hip", tmp_path)
    nn = _functions("nn_kernels.hip", tmp_path)
    checked = 0
    for name, lines in list(conv3.items()) + list(nn.items()):
        if not any("global_load_lds" in l for l in lines):
            continue
        waits = _barrier_waits(lines)
        assert waits, f"{name}: a DMA kernel without a barrier"
        if "lin_fk_kernel" in name or "lin_mp_kernel" in name or "attn_lds_kernel" in name:
            assert all(waits), f"{name}: a barrier of the DMA ring has no vmcnt wait in front of it: {waits}"
        else:
            # conv3: the item barrier and the zero-fill barrier publish no copies; the chunk-loop barrier (the last one) does
            assert waits[-1], f"{name}: the chunk-loop barrier has no vmcnt wait in front of it: {waits}"
        checked += 1
    assert checked >= 40, f"only {checked} DMA kernels found: the scan no longer sees the instantiations"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_direct_operand_kernels_issue_their_loads_before_the_first_mfma(tmp_path):
    """audio3_kernel / convs2d_kernel (conv7_mfma.hip, round 6) feed their MFMAs straight from global memory; what makes them faster
    than the LDS-staged launches they replaced is that ALL operand loads of a tile are in flight before its first MFMA (a
    sched_barrier between the phases).  Left to its own schedule hipcc pairs every few loads with their MFMAs - as many serial
    round trips, and exactly as slow as the old launches (profiles/r06_audio0_ab.txt) - without any test failing.  Likewise
    audio0_kernel's nine mel loads must be issued back to back, not as nine load / wait pairs."""
    funcs = _functions("conv7_mfma.hip", tmp_path)

    def loads_before_first(lines, what, load="global_load_dwordx4"):
        n = 0
        for l in lines:
            if what in l:
                return n
            if load in l:
                n += 1
        raise AssertionError(f"no {what} found")

    seen = 0
    for name, lines in funcs.items():
        if "audio3_kernel" in name:
            assert loads_before_first(lines, "v_mfma") >= 54, name           # 18 pixel operands + 36 weight operands
            seen += 1
        elif "convs2d_kernel" in name:
            cb = 2 if "ILi2E" in name else 1
            assert loads_before_first(lines, "v_mfma") >= 18 * cb, name      # 9 CB weight operands + 9 CB pixel operands of the first tile
            seen += 1
        elif "audio0_kernel" in name:
            first_wait = next(i for i, l in enumerate(lines) if "s_waitcnt" in l and "vmcnt" in l)
            issued = sum(1 for l in lines[:first_wait] if "global_load_dword " in l or "global_load_dword\t" in l or l.strip().startswith("global_load_dword v"))
            assert issued >= 9, f"{name}: {issued} mel loads in front of the first vmcnt wait"
            seen += 1
    assert seen == 4, f"{seen} of the 4 kernels found"
