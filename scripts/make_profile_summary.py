#!/usr/bin/env python3
"""Turn one scripts/gpu_job.sh profile output directory (gpurun_out/<tag>) into the committed summaries under
profiles/: per-kernel stats (rocprofv3 --kernel-trace --stats), per-launch timeline of one inference pass, and
the PMC-derived HBM traffic / MFMA utilisation of the conv kernels.

    python scripts/make_profile_summary.py gpurun_out/r03 profiles/r03 --name w2l --cmd "bench.py --steps 6 --warmup 2" [--all-kernels]

`--name` selects the workload's sub-directories (<name>_trace, <name>_pmc_fetch, ...) that `gpu_job.sh profile` writes; the outputs
are <dst_prefix>_<name>_kernel_stats.csv, _pmc.json, _pmc_per_kernel.csv (and _pass_timeline.txt for the Wav2Lip workloads).

HBM bytes follow MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of
the bytes of a wide coalesced stream (TCC_EA0_RDREQ tallied at 64 B for 128-B requests), so reads are doubled;
WRITE_SIZE is taken as reported (checked here against a layer whose output size is known exactly).
"""
import argparse
import collections
import json
import os
import re
import sqlite3


def is_layer_kernel(name):
    """Same predicate as bench.py's (tests/test_abi_and_host.py holds the two together): the kernels that run the conv / convT layers of a
    Wav2Lip pass.  Templated kernels are listed demangled ("conv3_kernel<...>"), the others mangled ("_ZN3ltk14convs2d_kernel...")."""
    return "conv" in name or "rowgemm" in name or "audio0_kernel" in name or "audio3_kernel" in name


def short(name):
    name = re.sub(r"\(.*$", "", name)
    return name.replace("void ", "").replace("ltk::", "")


def counters(path):
    if not os.path.exists(path):
        return {}
    db = sqlite3.connect(path)
    out = collections.defaultdict(dict)
    for k, c, v, n in db.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name"):
        out[short(k)][c] = (v, n)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("dst_prefix")
    ap.add_argument("--passes", type=int, default=0, help="passes in the profiled command; 0 = counted from the trace (Wav2Lip: launches of the fused "
                                                           "head kernel = one per pass; MuseTalk: gn_apply launches / 91)")
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--name", default="", help="workload prefix of the sub-directories (w2l, w2l256, mt, mtfp8); empty = round-2 layout")
    ap.add_argument("--cmd", default="bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-also --no-traffic", help="the profiled command (for the headers)")
    ap.add_argument("--all-kernels", action="store_true", help="PMC sums over every kernel of the run except the runtime's copy / fill blits (MuseTalk)")
    a = ap.parse_args()
    sub = (lambda d: os.path.join(a.src, f"{a.name}_{d}")) if a.name else (lambda d: os.path.join(a.src, d))
    if a.name:
        a.dst_prefix = f"{a.dst_prefix}_{a.name}"
    os.makedirs(os.path.dirname(a.dst_prefix) or ".", exist_ok=True)
    db = sqlite3.connect(os.path.join(sub("trace"), "r_results.db"))
    try:
        rows = db.execute("select name, start, end, grid_x, workgroup_x, lds_size, vgpr_count, sgpr_count, stream_id, grid_y, workgroup_y "
                          "from kernels order by start").fetchall()
    except sqlite3.OperationalError:
        rows = [r + (0, 1) for r in db.execute("select name, start, end, grid_x, workgroup_x, lds_size, vgpr_count, sgpr_count, stream_id "
                                               "from kernels order by start").fetchall()]
    stats = collections.OrderedDict()
    for r in rows:
        s = stats.setdefault(short(r[0]), [0, 0.0, 1e30, 0.0, r[6], r[7]])
        d = (r[2] - r[1]) / 1e3
        s[0] += 1; s[1] += d; s[2] = min(s[2], d); s[3] = max(s[3], d)
    tot = sum(s[1] for s in stats.values())
    if a.passes <= 0:
        heads = sum(v[0] for k, v in stats.items() if "conv3_head_kernel" in k)
        gn = sum(v[0] for k, v in stats.items() if "gn_apply_kernel" in k)
        fills = sum(v[0] for k, v in stats.items() if "gn_fill_kernel" in k)        # one per MuseTalk program run since the one-pass GroupNorm (round 6)
        a.passes = heads if heads else fills if fills else max(1, gn // 91)
    with open(a.dst_prefix + "_kernel_stats.csv", "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats -- python {a.cmd}\n")
        f.write("kernel,calls,total_us,avg_us,min_us,max_us,pct,vgpr,sgpr\n")
        for k, s in sorted(stats.items(), key=lambda kv: -kv[1][1]):
            f.write(f"\"{k}\",{s[0]},{s[1]:.1f},{s[1]/s[0]:.2f},{s[2]:.2f},{s[3]:.2f},{100*s[1]/tot:.2f},{s[4]},{s[5]}\n")
    # one inference pass = everything behind the previous pass's last kernel up to and including this pass's last kernel: the fused
    # head (conv3_head_kernel / head_kernel) for Wav2Lip, the frame writer (vae_post_kernel) for MuseTalk.  (Since round 5 a Wav2Lip
    # pass may carry the NEXT call's face encoder beside it - knob PREFETCH -, whose first launches go out in front of the mel pack:
    # delimiting by the end marker keeps them in the pass that issued them.)
    def is_end(name):
        return "head_kernel" in name or "vae_post_kernel" in name
    ends = [i for i, r in enumerate(rows) if is_end(r[0])]
    conv_us = 0.0
    with open(a.dst_prefix + "_pass_timeline.txt", "w") as f:
        f.write("# last inference pass of the profiled run: start_us dur_us stream grid lds_bytes kernel\n")
        if ends:
            last = ends[-1]
            # LIST_PASSES=2: the last TWO end-delimited segments (a staggered MuseTalk call ends twice: once per half-batch)
            back = 1 + int(os.environ.get("LIST_PASSES", "1"))
            first = ends[-back] + 1 if len(ends) >= back else 0
            # the pointer-table upload belongs to the pass it precedes; skip host-side blits between passes
            seg = [r for r in rows[first:last + 1] if "__amd_rocclr" not in r[0]]
            t0 = seg[0][1]
            pm = [r for r in seg if "pack_mel" in r[0] or "gather_latents" in r[0]]
            nfr = (pm[0][9] // max(pm[0][10], 1)) if pm and pm[0][9] else 0      # one grid row per frame
            f.write(f"# frames in the listed pass: {nfr if nfr else a.frames}\n")
            streams = sorted({r[8] for r in seg})
            for r in seg:
                n = short(r[0])
                f.write(f"{(r[1]-t0)/1e3:9.1f} {(r[2]-r[1])/1e3:8.1f} s{streams.index(r[8])} {r[3]//max(r[4],1):6d} {r[5]:7d} {n}\n")
                if a.all_kernels or is_layer_kernel(n):
                    conv_us += (r[2] - r[1]) / 1e3
            f.write(f"# first start .. last end: {(max(r[2] for r in seg)-t0)/1e3:.1f} us; sum of {'all' if a.all_kernels else 'conv'} kernels "
                    f"{conv_us:.1f} us; {len(seg)} launches on {len(streams)} stream(s)\n")

    fetch = counters(os.path.join(sub("pmc_fetch"), "r_results.db"))
    write = counters(os.path.join(sub("pmc_write"), "r_results.db"))
    sq = counters(os.path.join(sub("pmc_sq"), "r_results.db"))
    l2 = counters(os.path.join(sub("pmc_l2"), "r_results.db"))
    conv = [k for k in stats if (("__amd_rocclr" not in k) if a.all_kernels else is_layer_kernel(k))]
    blits = {k: stats[k][0] for k in stats if "__amd_rocclr" in k}
    rd = sum(fetch.get(k, {}).get("FETCH_SIZE", (0, 0))[0] for k in conv) * 1024 * 2
    wr = sum(write.get(k, {}).get("WRITE_SIZE", (0, 0))[0] for k in conv) * 1024
    mfma = sum(sq.get(k, {}).get("SQ_VALU_MFMA_BUSY_CYCLES", (0, 0))[0] for k in conv)
    gui = sum(l2.get(k, {}).get("GRBM_GUI_ACTIVE", (0, 0))[0] for k in conv)       # summed over the 8 XCDs
    hit = sum(l2.get(k, {}).get("TCC_HIT_sum", (0, 0))[0] for k in conv)
    miss = sum(l2.get(k, {}).get("TCC_MISS_sum", (0, 0))[0] for k in conv)
    wave = sum(sq.get(k, {}).get("SQ_WAVE_CYCLES", (0, 0))[0] for k in conv)
    summary = {
        "command": "python " + a.cmd,
        "kernels_counted": "all kernels of the run except the runtime's copy / fill blits" if a.all_kernels else "conv kernels",
        "runtime_blit_launches_in_run": blits,
        "passes_in_run": a.passes,
        "passes_note": "one pass = one launch sequence (<= one arena of frames); with session threads the coalesced calls differ in size, so "
                       "per-pass figures of a multi-session run are averages - bench.py's own roofline.traffic (fixed-size passes) is the reference figure",
        "frames_per_pass": a.frames,
        "hbm_read_bytes_per_pass": rd / a.passes,
        "hbm_write_bytes_per_pass": wr / a.passes,
        "hbm_bytes_per_pass": (rd + wr) / a.passes,
        "hbm_bytes_per_frame": (rd + wr) / a.passes / a.frames,
        "fetch_size_correction": "x2 (gfx950 FETCH_SIZE tallies 128-B requests at 64 B)",
        "mfma_busy_cycles_per_pass": mfma / a.passes,
        "mfma_util_conv_kernels": (mfma / 1024.0) / (gui / 8.0) if gui else None,
        "l2_hit_rate_conv_kernels": hit / (hit + miss) if hit + miss else None,
        "conv_kernel_us_last_pass": conv_us,
        "wave_cycles_split": {c: (sum(sq.get(k, {}).get(c, (0, 0))[0] for k in conv) / wave if wave else None)
                              for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")},
    }
    with open(a.dst_prefix + "_pmc.json", "w") as f:
        json.dump(summary, f, indent=1)
    with open(a.dst_prefix + "_pmc_per_kernel.csv", "w") as f:
        f.write("kernel,dispatches,FETCH_SIZE_KiB,WRITE_SIZE_KiB,MFMA_BUSY_CYCLES,GRBM_GUI_ACTIVE,TCC_HIT,TCC_MISS,LDS_BANK_CONFLICT,LDS_IDX_ACTIVE\n")
        for k in conv:
            g = lambda d, c: d.get(k, {}).get(c, (0, 0))[0]
            f.write(f"\"{k}\",{stats[k][0]},{g(fetch,'FETCH_SIZE'):.0f},{g(write,'WRITE_SIZE'):.0f},{g(sq,'SQ_VALU_MFMA_BUSY_CYCLES'):.0f},"
                    f"{g(l2,'GRBM_GUI_ACTIVE'):.0f},{g(l2,'TCC_HIT_sum'):.0f},{g(l2,'TCC_MISS_sum'):.0f},{g(sq,'SQ_LDS_BANK_CONFLICT'):.0f},{g(sq,'SQ_LDS_IDX_ACTIVE'):.0f}\n")
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
