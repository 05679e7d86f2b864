#!/usr/bin/env python3
"""inferfps bench of the MI355X-native Wav2Lip-256 render hot path.

A "step" is one `LipReal.inference_batch`-equivalent pass (bank gather + mask +
pack, the 55 conv/convT layers, sigmoid*255 + uint8 truncation) over one batch
of B=16 frames per session, with the avatar bank, the weights and the mel
windows already resident in HBM (avatars/base_avatar.py:364-373 defines
inferfps as frames / wall time of inference_batch).  BASELINE.json configs[1]:
wav2lip256, 1 session, 16-frame batch, fp16 on 1x MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--sessions S] [--batch B]

N>1 is launched by torch.distributed.run (one rank per GPU, RCCL only for the
barrier / max-over-ranks reduction; sessions are independent, so the data path
has no collective).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MACS_PER_FRAME = 27_788_599_296       # SURVEY.md Appendix A (conv + convT + head)
HEAD_MACS = 32 * 3 * 65536
PEAK_F16_TFLOPS = 2500.0              # MI355X_MICROARCH.md: dense fp16/bf16 MFMA


def cpu_baseline(batch: int, budget_s: float = 20.0):
    """The oracle restatement of LipReal.inference_batch (fp32, torch CPU) on the
    host cores, bounded sample."""
    import numpy as np
    import torch
    from livetalking_amd import synth
    from oracle import mel_oracle, plugin_oracle   # the checker, timed here as the CPU baseline only
    torch.set_num_threads(min(64, os.cpu_count() or 1))   # more threads only add contention on this model size
    sd = {k: torch.from_numpy(v) for k, v in synth.wav2lip_state_dict(1234).items()}
    frames, faces, coords = synth.wav2lip_avatar(n_frames=8, full_hw=(360, 640), box=160, seed=0)
    audio = synth.synthetic_audio(2.0)
    feats = mel_oracle.mel_chunks(audio[: (20 + 2 * batch) * 320], 20 + 2 * batch)
    plugin_oracle.inference_batch(sd, faces, 0, 1, feats[:1])  # warm-up (B=1)
    n, t_total = 0, 0.0
    while True:
        t0 = time.perf_counter()
        plugin_oracle.inference_batch(sd, faces, n * batch, batch, feats)
        t_total += time.perf_counter() - t0
        n += 1
        if t_total >= budget_s * 0.5 or n >= 4:
            break
    return {"value": round(n * batch / t_total, 3), "unit": "frames/s", "cores": torch.get_num_threads(),
            "kind": "port", "sample": f"{n} x inference_batch(B={batch}) fp32 torch-CPU oracle, seeded synthetic weights/bank/audio"}


def cpu_baseline_musetalk(budget_s: float = 25.0):
    """The oracle restatement of MuseReal.inference_batch (fp32, torch CPU) on the host cores, bounded sample."""
    import torch
    from livetalking_amd import synth
    from oracle import musetalk_oracle as M          # the checker, timed here as the CPU baseline only
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    usd = {k: torch.from_numpy(v) for k, v in synth.musetalk_unet_state_dict().items()}
    vsd = {k: torch.from_numpy(v) for k, v in synth.vae_decoder_state_dict().items()}
    lats = [torch.from_numpy(x) for x in synth.musetalk_latents(2)]
    feats = synth.musetalk_whisper_feats(1)
    n, t_total = 0, 0.0
    with torch.no_grad():
        while True:
            t0 = time.perf_counter()
            M.inference_batch(usd, vsd, lats, n, 1, feats)
            t_total += time.perf_counter() - t0
            n += 1
            if t_total >= budget_s * 0.5 or n >= 3:
                break
    return {"value": round(n / t_total, 3), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} x inference_batch(B=1) fp32 torch-CPU oracle (U-Net + VAE decoder), seeded synthetic weights"}


def main_musetalk(args):
    """BASELINE.json configs[2]: MuseTalk (Whisper audio feat + U-Net + VAE decoder), 1 session, 1 GPU.  A step =
    one MuseReal.inference_batch-equivalent pass (latent gather + PE + U-Net + VAE decode + uint8 BGR) over
    sessions x batch frames with latents, weights and whisper chunks resident in HBM; the Whisper step
    (run_step's work, outside inferfps in the reference too) is timed separately."""
    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    from livetalking_amd import synth
    from livetalking_amd.engine import Engine

    S, B = args.sessions, args.batch
    fps_step = S * B
    eng = Engine(local_rank)
    eng.load_musetalk(synth.musetalk_unet_state_dict(), synth.vae_decoder_state_dict(), max_frames=min(fps_step, 64), fp8=args.fp8)
    macs_all, macs_fp8 = eng.musetalk_info()
    n = 8
    lats = synth.musetalk_latents(n)
    frames, _, _ = synth.wav2lip_avatar(n_frames=n, full_hw=(720, 1280), box=320, seed=0)
    coords = [(480, 200, 800, 520)] * n
    crops = [(400, 120, 880, 600)] * n
    masks = [np.full((480, 480, 3), 128, np.uint8)] * n
    aid = eng.register_musetalk_avatar(lats, frames, coords, masks, crops)
    d_feat = torch.from_numpy(synth.musetalk_whisper_feats(fps_step)).cuda().reshape(S, B, 50, 384)
    d_pred = torch.zeros(S, B, 256, 256, 3, dtype=torch.uint8, device="cuda")

    def step(i):
        eng.musetalk_infer([(aid, i * B + 3 * s, B, d_feat[s].data_ptr(), d_pred[s].data_ptr()) for s in range(S)])

    for i in range(args.warmup):
        step(i)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    value = world * args.steps * fps_step / elapsed
    nt = min(fps_step, 64)
    ms, macs = eng.musetalk_time(nt, 3)
    achieved = 2.0 * macs / (ms * 1e-3) / 1e12
    # WhisperASR.run_step's feature work (log-mel + whisper-tiny encoder + chunk slicing) for one session's 0.64-s step: the
    # reference runs it on the render thread, outside inferfps (base_avatar.py:364-373 times inference_batch only); reported
    # beside it because BASELINE configs[2] names it
    eng.load_whisper(synth.whisper_encoder_state_dict())
    pcm = synth.synthetic_audio(2.0)[: (20 + 2 * B) * 320]
    d_chunks = torch.zeros(B, 50, 384, dtype=torch.float32, device="cuda")
    for _ in range(2):
        eng.whisper_step(pcm, B, first_row=10, d_out_ptr=d_chunks.data_ptr())
    torch.cuda.synchronize()
    tw = time.perf_counter()
    for _ in range(5):
        eng.whisper_step(pcm, B, first_row=10, d_out_ptr=d_chunks.data_ptr())
    torch.cuda.synchronize()
    whisper_ms = (time.perf_counter() - tw) / 5 * 1e3
    if rank == 0:
        out = {"metric": "inferfps", "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "fp8(e4m3)+f16" if args.fp8 else "f16", "data": "synthetic",
               "config": {"workload": f"musetalk (U-Net + VAE decoder), {S} session(s)/GPU, {B}-frame batch, " +
                                      (f"fp8 e4m3 operands on the resnet 3x3 convs ({macs_fp8 / macs_all:.0%} of the MACs), fp16 elsewhere, fp32 accumulate"
                                       if args.fp8 else "fp16 activations / fp32 accumulate"),
                          "sessions_per_gpu": S, "batch": B, "frames_per_step_per_gpu": fps_step,
                          "parallelism": f"session-sharded x{world} (no collective)"},
               "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                            "frac": round(achieved / PEAK_F16_TFLOPS, 5), "traffic": None,
                            "kernel": "conv3_kernel / conv_mfma_kernel (U-Net + VAE conv and linear layers; attention excluded from the flop count)",
                            "pass_ms": round(ms, 4), "flops_per_frame": 2.0 * macs / nt}}
        out["whisper"] = {"ms_per_session_step": round(whisper_ms, 3), "frames_per_step": B,
                          "note": "log-mel + whisper-tiny encoder (1500 tokens) + chunk slicing, host PCM in, device features out; not in `value`",
                          "fps_incl_whisper": round(fps_step / (elapsed / args.steps + S * whisper_ms * 1e-3), 2)}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline_musetalk()
        print(json.dumps(out), flush=True)
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--sessions", type=int, default=1, help="sessions coalesced per launch on each GPU")
    ap.add_argument("--batch", type=int, default=16, help="frames per session per step (opt.batch_size)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--paced", type=int, default=0, help="wav2lip: after the timed run, N periods of B/25 s with all sessions paced at 25 fps")
    ap.add_argument("--fp8", action="store_true", help="musetalk: BASELINE configs[4] fp8 conv path (non-scaled fp8 MFMA = the fp16 MFMA rate, so the roofline peak stays 2.5 PF)")
    ap.add_argument("--model", choices=("wav2lip", "musetalk"), default="wav2lip",
                    help="wav2lip = BASELINE.json configs[1] (default, the driver's line); musetalk = configs[2]")
    args = ap.parse_args()
    if args.model == "musetalk":
        return main_musetalk(args)

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local_rank)

    from livetalking_amd.engine import Engine
    from livetalking_amd import synth  # seeded synthetic input generators (product-side; nothing from oracle/)

    S, B = args.sessions, args.batch
    frames_per_step = S * B
    if frames_per_step > 256:
        os.environ.setdefault("LTK_MICROBATCH", "256")     # activation arena for 256 frames; larger steps run as micro-batches
    eng = Engine(local_rank)
    call_sessions = max(1, min(S, 4096 // B))              # ltk_wav2lip_infer takes up to 4096 frames per call
    eng.load_wav2lip(synth.wav2lip_state_dict(1234), max_frames=call_sessions * B)
    frames, faces, coords = synth.wav2lip_avatar(n_frames=32, full_hw=(720, 1280), box=320, seed=0)
    aid = eng.register_avatar(faces, frames, coords)
    # mel windows resident in HBM: one (B,80,16) block per session, made by the HIP mel kernel
    audio = synth.synthetic_audio(4.0)
    n_chunks = 20 + 2 * B
    starts = [int(16 + i * 3.2) for i in range(B)]
    d_mel = torch.zeros(S, B, 80, 16, dtype=torch.float32, device="cuda")
    for s in range(S):
        off = (s * 977) % (len(audio) - n_chunks * 320)
        eng.mel_step(audio[off: off + n_chunks * 320], starts, d_mel[s].data_ptr())
    d_pred = torch.zeros(S, B, 256, 256, 3, dtype=torch.uint8, device="cuda")

    def step(i):
        reqs = [(aid, i * B + 7 * s, B, d_mel[s].data_ptr(), d_pred[s].data_ptr()) for s in range(S)]
        for j in range(0, S, call_sessions):
            eng.wav2lip_infer(reqs[j:j + call_sessions])

    for i in range(args.warmup):
        step(i)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    total_frames = world * args.steps * frames_per_step
    value = total_frames / elapsed

    # --paced: SURVEY.md 8(d)'s second accounting.  Every session asks for its next B frames once per B/25 s (what a
    # 25 fps render loop does); all S requests of a period go down as one coalesced call.  A period's latency is the
    # time from its due time to the last frame being ready: the session count is sustainable if it stays below the
    # period.  Not part of the timed region above.
    paced = None
    if args.paced > 0:
        period = B / 25.0
        lat = []
        tp0 = time.perf_counter() + 0.05
        for i in range(args.paced):
            due = tp0 + i * period
            while time.perf_counter() < due:
                time.sleep(0.0005)
            step(args.warmup + args.steps + i)          # returns when the frames are ready (ltk_wav2lip_infer is synchronous)
            lat.append(time.perf_counter() - due)
        paced = {"sessions": S, "fps_per_session": 25, "periods": args.paced, "period_ms": period * 1e3,
                 "latency_ms_mean": round(1e3 * sum(lat) / len(lat), 2), "latency_ms_max": round(1e3 * max(lat), 2),
                 "sustained": bool(max(lat) < period)}

    # dominant kernel family (conv3_kernel / conv_mfma_kernel: the 54 conv layers of one pass): HIP events on
    # the engine's own stream around the conv stack only (no gather/pack, no head), averaged over 10 passes
    # of the same workload.  One "launch" below = one pass of the conv stack over frames_per_step frames.
    conv_ms, conv_macs = eng.time_convs(min(frames_per_step, call_sessions * B), 10)
    achieved = 2.0 * conv_macs / (conv_ms * 1e-3) / 1e12
    # HBM bytes per pass from the committed rocprofv3 PMC summary of this same command
    # (scripts/gpu_profile.sh -> scripts/make_profile_summary.py): separate --pmc passes, FETCH_SIZE doubled
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc.json")) as f:
            pm = json.load(f)
        if int(pm.get("frames_per_pass", -1)) == frames_per_step:
            traffic = float(pm["hbm_bytes_per_pass"])
    except Exception:  # noqa: BLE001 - the summary is optional
        traffic = None

    if rank == 0:
        out = {
            "metric": "inferfps",
            "value": round(value, 2),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16",
            "data": "synthetic",
            "config": {"workload": f"wav2lip256, {S} session(s)/GPU, {B}-frame batch, fp16 activations / fp32 accumulate",
                       "sessions_per_gpu": S, "batch": B, "frames_per_step_per_gpu": frames_per_step,
                       "parallelism": f"session-sharded x{world} (no collective)"},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_F16_TFLOPS, 5), "traffic": traffic,
                         "traffic_unit": "HBM bytes per pass (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, profiles/r01_pmc.json)",
                         "kernel": "conv3_kernel + conv_mfma_kernel (the 54 conv/convT layers = one pass)",
                         "conv_stack_ms": round(conv_ms, 4), "flops_per_frame": 2 * (MACS_PER_FRAME - HEAD_MACS)},
        }
        # BASELINE.json's second figure: sessions a GPU sustains at 25 fps each.  A step of S coalesced sessions must finish
        # within the 0.64 s its 16 frames last (base_avatar.py:364-373 accounting), so the bound is throughput / 25 as long
        # as the step latency stays below that.
        step_ms = elapsed / args.steps * 1e3
        out["sessions_25fps"] = {"per_gpu": int(value / world // 25), "step_latency_ms": round(step_ms, 3),
                                 "latency_budget_ms": 1000.0 * B / 25, "note": "unpaced saturating rate / 25 fps"}
        if paced is not None:
            out["paced"] = paced
        if not args.no_cpu_baseline and world == 1:         # the CPU baseline is timed at N=1 only
            out["cpu_baseline"] = cpu_baseline(B)
        print(json.dumps(out), flush=True)
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
