#!/usr/bin/env python3
"""The CPU baseline of bench.py in BOTH kinds on this host: `reference` (the reference's own LipReal.inference_batch from a LiveTalking
checkout) and `port` (the oracle restatement, what the GPU box times because it has no checkout), each in its own process, twice,
interleaved.  Writes a JSON record (default profiles/r05_cpu_baseline_build_container.json).  Build container only.

    python scripts/cpu_baseline_compare.py [out.json]
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = "import json, bench; print(json.dumps(bench.cpu_baseline(16)))"


def run(kind):
    env = dict(os.environ, LTK_ALLOW_STANDIN="1")
    if kind == "port":
        env["LTK_CPU_BASELINE_KIND"] = "port"
    r = subprocess.run([sys.executable, "-c", CODE], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1800)
    for line in reversed(r.stdout.strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise SystemExit(f"{kind}: no result: {r.stderr[-500:]}")


def quick():
    """In-process, interleaved, B = 4 on the small bank: port / reference wall-time ratio of up to three attempts (CPU test)."""
    import time
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    os.environ.setdefault("LTK_ALLOW_STANDIN", "1")
    import synth_inputs as synth
    from oracle import mel_oracle, plugin_oracle, ref_loop
    ref = os.environ.get("LTK_REFERENCE", "/root/reference")
    cwd = os.getcwd()
    ref_loop.enter_reference(ref)
    import avatars.wav2lip_avatar as ref_plugin
    from avatars.wav2lip.models import Wav2Lip
    os.chdir(cwd)
    sd = {k: torch.from_numpy(v) for k, v in synth.wav2lip_state_dict(1234).items()}
    frames, faces, coords = synth.wav2lip_avatar(n_frames=8, full_hw=(360, 640), box=160, seed=0)
    net = Wav2Lip().eval()
    net.load_state_dict(sd)
    lip = ref_plugin.LipReal.__new__(ref_plugin.LipReal)
    B = 4
    lip.model, lip.batch_size = net, B
    lip.frame_list_cycle, lip.face_list_cycle, lip.coord_list_cycle = frames, faces, coords
    feats = mel_oracle.mel_chunks(synth.synthetic_audio(2.0)[: (20 + 2 * B) * 320], 20 + 2 * B)
    calls = {"reference": lambda i: lip.inference_batch(i, feats), "port": lambda i: plugin_oracle.inference_batch(sd, faces, i, B, feats)}
    diff = float(np.abs(np.asarray(calls["reference"](0)) - np.asarray(calls["port"](0))).max())
    ratios = []
    for attempt in range(3):
        t = {"reference": [], "port": []}
        for rnd in range(3):
            for k in ("reference", "port"):
                t0 = time.perf_counter()
                calls[k](1 + rnd)
                t[k].append(time.perf_counter() - t0)
        ratios.append(round(float(np.median(t["port"]) / np.median(t["reference"])), 4))
        if abs(ratios[-1] - 1.0) <= 0.15:
            break
    print(json.dumps({"port_over_reference": ratios, "max_abs_frame_diff": diff, "batch": B}))


def main():
    if "--quick" in sys.argv:
        return quick()
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r05_cpu_baseline_build_container.json")
    runs = [run(k) for k in ("reference", "port", "reference", "port")]
    assert runs[0]["kind"] == "reference" and runs[1]["kind"] == "port", "needs a LiveTalking checkout (LTK_REFERENCE)"
    ref16 = [r["value"] for r in runs if r["kind"] == "reference"]
    port16 = [r["value"] for r in runs if r["kind"] == "port"]
    ref1 = [r["b1"]["value"] for r in runs if r["kind"] == "reference"]
    port1 = [r["b1"]["value"] for r in runs if r["kind"] == "port"]
    rec = {"host": "build container", "cores": runs[0]["cores"], "cpu_count": os.cpu_count(),
           "reference_fps_b16": ref16, "port_fps_b16": port16, "reference_fps_b1": ref1, "port_fps_b1": port1,
           "port_over_reference_b16": round(sum(port16) / sum(ref16), 4), "port_over_reference_b1": round(sum(port1) / sum(ref1), 4),
           "runs": runs,
           "note": "bench.py's cpu_baseline leg, kind reference = the reference's own LipReal.inference_batch (avatars/wav2lip_avatar.py:116-139), "
                   "kind port = oracle/plugin_oracle.py, each in its own process, interleaved; the GPU box has no checkout and reports kind port"}
    with open(out, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps({k: rec[k] for k in ("cores", "reference_fps_b16", "port_fps_b16", "reference_fps_b1", "port_fps_b1",
                                          "port_over_reference_b16", "port_over_reference_b1")}))


if __name__ == "__main__":
    main()
