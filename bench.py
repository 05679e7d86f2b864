#!/usr/bin/env python3
"""inferfps bench of the MI355X-native LiveTalking render hot path (BASELINE.json: "inferfps/GPU + max concurrent
25 fps sessions, wav2lip256 & MuseTalk").

A "step" is one `inference_batch` call per session (avatars/base_avatar.py:366) - bank gather + mask + pack, the conv
stack, head, uint8 frames - issued through the PLUGIN surface (`LipReal.inference_batch` / `MuseReal.inference_batch`:
torch.empty of the outputs, scheduler, ctypes marshalling included), with the avatar bank, the weights and the audio
features already resident in HBM.  inferfps = frames / wall time of those calls (base_avatar.py:364-373).  With several
sessions every session has its own thread and issues its K calls back to back, like the reference's per-session inference
threads; the threads meet only before the first and after the last call.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--sessions S] [--batch B] [--model wav2lip|musetalk] [--fp8]

Default (N=1): the timed line is BASELINE.json configs[1] (wav2lip256, 1 session, 16-frame batch, fp16).  Rank 0 then
adds, OUTSIDE the timed region (each in its own subprocess, so `ms_per_step x steps` stays what was timed):
  also[]        configs[3]'s per-GPU share (16 wav2lip sessions on one GPU, saturating and paced at 25 fps),
                configs[2] (MuseTalk + Whisper step) and configs[4]'s per-GPU share (4 MuseTalk sessions) in fp16 AND with the
                fp8 conv path, each with its own PMC `roofline.traffic`
  paced         the largest number of 25-fps wav2lip sessions one GPU sustains with the frames left on the device
                (bisection, engine level: "kernel capacity")
  delivered     the same with every session's 16 composited 720p frames copied to the host per period, through the plugin
                (inference_batch + paste_back_frame, one thread per session): what a deployment can actually serve;
                .bgr24 through paste_back_frame, .i420 through the plugin's opt.egress path (composite + watermark + BGR->I420 on the GPU)
  cpu_baseline  the reference's LipReal.inference_batch on the host cores (kind "reference" when a LiveTalking checkout
                is importable, else the oracle port), B=16 and B=1 (configs[0]), median of 5
  roofline.traffic  HBM bytes per pass from two rocprofv3 --pmc passes (FETCH_SIZE x2, WRITE_SIZE) of the conv stack
  pcie_inclusive    the same single session with host mel in and every composited frame copied back (paste_back_frame)

N>1: sessions are independent (app.py:62-63,99), so rank r owns its own engine, bank replica and sessions: no collective
on the data path, xGMI unused.  Launched either by the driver (`python -m torch.distributed.run ... bench.py --gpus N`:
RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* in the environment) or directly (`python bench.py --gpus N` spawns the N ranks
itself).  Ranks meet on a gloo (CPU) barrier on both sides of the timed region and rank 0 takes the max elapsed time;
`--dry-ranks` runs that protocol without touching a GPU (CPU test of the launcher).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("LTK_ALLOW_STANDIN", "1")     # headless: stand-ins of the reference's BaseAvatar/BaseASR (hostshim.py)

MACS_PER_FRAME = 27_788_599_296       # SURVEY.md Appendix A (54 conv/convT layers + the 1x1 head)
PEAK_F16_TFLOPS = 2500.0              # MI355X_MICROARCH.md: dense fp16/bf16 MFMA
PEAK_FP8_TFLOPS = 5000.0              # MI355X_MICROARCH.md: dense fp8 MFMA (MX-scaled instruction)
REF_ROOT = os.environ.get("LTK_REFERENCE", "/root/reference")
BANK_FRAMES = 250                     # avatar bank of the bench (740 MB of 720p frames + 49 MB of face crops per GPU)


# ---------------------------------------------------------------------------------------------------------------
# ranks
# ---------------------------------------------------------------------------------------------------------------
def rank_env():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def rank_device(local_rank: int) -> int:
    """Rank r renders on GPU r.  LTK_RANK_DEVICES="0,0" (explicit, test-only override) maps local ranks onto the listed GPUs
    instead - several ranks on ONE GPU exercise the real multi-rank path (engine per rank, gloo barriers around a GPU step) on a
    1-GPU box; without it a rank whose GPU does not exist fails loudly."""
    m = os.environ.get("LTK_RANK_DEVICES", "").strip()
    if not m:
        return local_rank
    devs = [int(v) for v in m.split(",") if v.strip() != ""]
    if local_rank >= len(devs):
        raise SystemExit(f"bench.py: LTK_RANK_DEVICES={m!r} names {len(devs)} ranks, local rank {local_rank} has no GPU")
    return devs[local_rank]


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(args, argv) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks (what torch.distributed.run would do), relay
    rank 0's line.  Fails loudly when the box has fewer GPUs than ranks."""
    n = args.gpus
    if not args.dry_ranks:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        need = 1 + max(rank_device(r) for r in range(n))
        if have < need:
            print(f"bench.py: --gpus {n} needs {need} visible GPUs, this box has {have}", file=sys.stderr)
            return 2
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), LTK_DEVICE=str(rank_device(r)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out, _ = procs[0].communicate()
    rc = procs[0].returncode
    for p in procs[1:]:
        rc = p.wait() or rc
    sys.stdout.write(out)
    sys.stdout.flush()
    return rc


class Ranks:
    """Barrier and max-over-ranks on the CPU (gloo): the data path has no collective to share a communicator with."""

    def __init__(self):
        self.rank, self.world, self.local_rank = rank_env()
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
            self.dist = dist

    def barrier(self, sync_gpu=True):
        if sync_gpu:
            import torch
            torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()

    def gather_max(self, value: float):
        """(max over ranks, list of every rank's value)."""
        if self.dist is None:
            return value, [value]
        import torch
        t = torch.tensor([value], dtype=torch.float64)
        allv = [torch.zeros(1, dtype=torch.float64) for _ in range(self.world)]
        self.dist.all_gather(allv, t)
        vals = [float(v.item()) for v in allv]
        return max(vals), vals

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------------
# session drivers (plugin surface)
# ---------------------------------------------------------------------------------------------------------------
class SessionThreads:
    """S sessions, each with its own thread calling `inference_batch` once per step - the reference's concurrency model
    (one inference thread per session, base_avatar.py:475-481).  S == 1 runs in the caller's thread."""

    def __init__(self, sessions, feats, stride):
        self.sessions, self.feats, self.stride = sessions, feats, stride
        self.S = len(sessions)
        self.busy = [0.0] * self.S          # per session: seconds inside inference_batch (the reference's counttime)
        self.frames = [0] * self.S
        self._step = 0
        self._nsteps = 1
        self._err = None
        if self.S > 1:
            self._go = threading.Barrier(self.S + 1)
            self._done = threading.Barrier(self.S + 1)
            self._stop = False
            self._due = None
            self._stagger = 0.0             # paced: session i asks i * _stagger seconds after the period starts
            self._threads = [threading.Thread(target=self._run, args=(i,), daemon=True) for i in range(self.S)]
            for t in self._threads:
                t.start()

    def _one(self, i, step):
        s = self.sessions[i]
        t = time.perf_counter()
        pred = s.inference_batch(step * s.batch_size + self.stride * i, self.feats[i])
        self.busy[i] += time.perf_counter() - t
        self.frames[i] += len(pred)

    def _run(self, i):
        while True:
            self._go.wait()
            if self._stop:
                return
            try:
                if self._due is not None:                      # paced: every session asks at its own due time
                    while time.perf_counter() < self._due + i * self._stagger:
                        time.sleep(0.0005)
                for st in range(self._nsteps):                 # free-running: no hand-shake with the other sessions between steps
                    self._one(i, self._step + st)
            except Exception as ex:  # noqa: BLE001
                self._err = ex
            self._done.wait()

    def step(self, step, due=None, nsteps=1):
        """`nsteps` consecutive inference_batch calls per session.  With several sessions every session's thread runs its
        calls back to back on its own, as the reference's per-session inference threads do (base_avatar.py:326-381): the
        threads meet only before the first and after the last call."""
        if self.S == 1:
            for st in range(nsteps):
                self._one(0, step + st)
            return
        self._step, self._due, self._nsteps = step, due, nsteps
        self._go.wait()
        self._done.wait()
        if self._err is not None:
            raise self._err

    def close(self):
        if self.S > 1:
            self._stop = True
            self._go.wait()


def paced_sessions(drv: SessionThreads, first_step: int, periods: int, B: int, stagger: bool = False):
    """Every session asks for its next B frames once per B/25 s (what a 25 fps render loop does).  A period is met when
    the last session's frames are ready before the next period starts.  `stagger`: the sessions' request times are spread
    evenly over the period (sessions that were started at different moments - the deployment case - instead of all at
    once): every request is then a call of its own, and `call_ms_mean` is the time a session waits inside inference_batch."""
    period = B / 25.0
    drv.busy = [0.0] * drv.S
    drv.frames = [0] * drv.S
    if drv.S > 1:
        drv._stagger = period / drv.S if stagger else 0.0
    lat = []
    t0 = time.perf_counter() + 0.05
    for p in range(periods):
        due = t0 + p * period
        if drv.S == 1:
            while time.perf_counter() < due:
                time.sleep(0.0005)
        drv.step(first_step + p, due=due)
        lat.append(time.perf_counter() - due - (drv._stagger * (drv.S - 1) if drv.S > 1 else 0.0))
    if drv.S > 1:
        drv._stagger = 0.0
    per_session = [f / b if b > 0 else 0.0 for f, b in zip(drv.frames, drv.busy)]
    return {"sessions": drv.S, "fps_per_session_required": 25, "periods": periods, "period_ms": period * 1e3, "staggered": bool(stagger),
            "call_ms_mean": round(1e3 * sum(drv.busy) / max(1, drv.S * periods), 4),
            "latency_ms_mean": round(1e3 * sum(lat) / len(lat), 2), "latency_ms_max": round(1e3 * max(lat), 2),
            "inferfps_per_session_min": round(min(per_session), 1), "inferfps_per_session_mean": round(sum(per_session) / len(per_session), 1),
            "sustained": bool(max(lat) < period and min(per_session) >= 25.0),
            "note": "inferfps per session = frames / time inside inference_batch (base_avatar.py:364-373)"}


# ---------------------------------------------------------------------------------------------------------------
# workloads
# ---------------------------------------------------------------------------------------------------------------
def run_wav2lip(args, ranks: Ranks):
    import argparse as ap
    import torch
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    dev = rank_device(ranks.local_rank)
    if dev >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {ranks.rank} needs GPU {dev}, this box has {torch.cuda.device_count()}")
    torch.cuda.set_device(dev)
    os.environ["LTK_DEVICE"] = str(dev)
    import livetalking_amd.avatars.wav2lip_avatar as plugin
    import synth_inputs as synth  # seeded synthetic input generators (repo root; nothing from oracle/)

    face_cache = os.environ.get("LTK_FACE_CACHE", "0") not in ("", "0")
    if face_cache and not args.sub:
        raise SystemExit("bench.py: LTK_FACE_CACHE is an opt-in deployment mode (the face encoder's outputs are cached per bank frame); "
                         "the timed line never uses it - it is reported on its own also[] entry")
    S, B = args.sessions, args.batch
    frames_per_step = S * B
    if frames_per_step > 4096:
        raise SystemExit("bench.py: at most 4096 frames per step (use --paced-capacity for the session capacity)")
    if frames_per_step > 256:
        os.environ.setdefault("LTK_MICROBATCH", "256")     # activation arena for 256 frames; larger steps run as micro-batches
    model = plugin.load_model(None, state_dict=synth.wav2lip_state_dict(1234), max_frames=frames_per_step, device=dev)
    eng = model.engine
    plugin.warm_up(B, model, 256)
    avatar = synth.wav2lip_bank(n_frames=BANK_FRAMES, full_hw=(720, 1280), box=320, seed=0)     # 250 frames, 720p, ~320-px boxes
    opt = ap.Namespace(fps=25, batch_size=B, l=10, r=10, sessionid=0)
    sessions = []
    for s in range(S):
        o = ap.Namespace(**vars(opt))
        o.sessionid = s
        sessions.append(plugin.LipReal(o, model, avatar))
    # mel windows resident in HBM: one (B,80,16) block per session, made by the HIP mel kernel
    audio = synth.synthetic_audio(4.0)
    n_chunks = 20 + 2 * B
    starts = [int(16 + i * 3.2) for i in range(B)]
    d_mel = torch.zeros(S, B, 80, 16, dtype=torch.float32, device="cuda")
    for s in range(S):
        off = (s * 977) % (len(audio) - n_chunks * 320)
        eng.mel_step(audio[off: off + n_chunks * 320], starts, d_mel[s].data_ptr())
    drv = SessionThreads(sessions, [d_mel[s] for s in range(S)], stride=7)

    # untimed priming in front of the W warm-up steps: a frame count's pass is captured as a hipGraph the SECOND time each of its
    # launch variants is seen (knob PREFETCH has three per frame count), and a capture inside the timed region would be timed;
    # the steps walk the bank in the order a session does, so that priming, warm-up and timed steps are one unbroken sequence
    PRIME = 8
    for i in range(PRIME):
        drv.step(i)
    for i in range(args.warmup):
        drv.step(PRIME + i)
    ranks.barrier()
    t0 = time.perf_counter()
    drv.step(PRIME + args.warmup, nsteps=args.steps)
    torch.cuda.synchronize()
    own = time.perf_counter() - t0                     # this rank's own time (reported per rank)
    ranks.barrier()
    elapsed = time.perf_counter() - t0
    elapsed_max, _ = ranks.gather_max(elapsed)
    _, per_rank = ranks.gather_max(own)
    value = ranks.world * args.steps * frames_per_step / elapsed_max

    # the sustained twin of the timed line: the SAME loop for >= `--sustain` seconds right behind the timed region (the first
    # launches of a process run up to ~20 % faster than the steady thermal state, and the default timed region is only ~30 ms)
    sustained = None
    if args.sustain > 0:
        n_sus = max(args.steps, int(args.sustain / max(elapsed_max / args.steps, 1e-6)) + 1)
        ranks.barrier()
        ts = time.perf_counter()
        drv.step(PRIME + args.warmup + args.steps, nsteps=n_sus)
        torch.cuda.synchronize()
        ranks.barrier()
        sus_max, _ = ranks.gather_max(time.perf_counter() - ts)
        sustained = {"value": round(ranks.world * n_sus * frames_per_step / sus_max, 2), "unit": "frames/s", "steps": n_sus,
                     "seconds": round(sus_max, 3), "ms_per_step": round(sus_max / n_sus * 1e3, 4),
                     "note": "same loop as the timed region, run for >= --sustain seconds directly behind it; not `value`"}

    paced = paced_sessions(drv, PRIME + args.warmup + args.steps + 4096, args.paced, B, stagger=args.paced_stagger) if args.paced > 0 else None
    sched = dict(sessions[0]._sched.stats)
    # PCIe-inclusive rate of one session (outside the timed region): mel windows start on the HOST (the reference's ASR hands
    # numpy arrays over, mel.py:34-67) and every frame comes back composited into its full frame as a host array, as
    # LipReal.paste_back_frame returns it (wav2lip_avatar.py:141-147): 82 KB up and B x H x W x 3 bytes down per step.
    pcie = None
    if ranks.rank == 0 and S == 1:
        import numpy as np
        from livetalking_amd.hostshim import mirror_index
        s0 = sessions[0]
        host_mel = [np.ascontiguousarray(m) for m in d_mel[0].cpu().numpy()]
        nloop = 10
        for it in range(nloop + 2):
            if it == 2:
                torch.cuda.synchronize()
                tp = time.perf_counter()
            pred = s0.inference_batch(it * B, host_mel)
            for i in range(B):
                s0.paste_back_frame(pred[i], mirror_index(BANK_FRAMES, it * B + i))
        dtp = time.perf_counter() - tp
        h, w = avatar[0][0].shape[:2]
        pcie = {"value": round(nloop * B / dtp, 1), "unit": "frames/s", "ms_per_step": round(dtp / nloop * 1e3, 3),
                "bytes_down_per_frame": int(h * w * 3),
                "note": "one session thread: host mel in, inference_batch, then paste_back_frame for each of the B frames "
                        "(B composites on the device, one pinned device-to-host copy per batch); not `value`"}
    drv.close()
    # dominant kernel family (conv3_kernel / conv_mfma_kernel: the 54 conv layers of one pass, the bank gather fused into the first,
    # the head into the last): HIP events on the engine's compute stream around the device side of one ltk_wav2lip_infer pass exactly
    # as that call enqueues it (mel pack + conv stack, replayed from the captured hipGraph under knob GRAPH), averaged over 10 passes
    # of the same workload.  One "launch" below = one pass over min(frames_per_step, 256) frames.
    nf_pass = min(frames_per_step, 256)
    graphs_timed_run = eng.graph_count()            # > 0: the timed inference_batch calls above ran from captured graphs
    pf_stats = eng.prefetch_stats()                 # knob PREFETCH: how the session calls above actually ran
    conv_ms, conv_macs = eng.time_convs(nf_pass, 10)
    achieved = 2.0 * conv_macs / (conv_ms * 1e-3) / 1e12
    # the same pass with every call running the whole network on its own (knob PREFETCH off: rounds 1-4), for comparison
    conv_ms_whole = None
    if pf_stats["issued"] > 0 and not args.no_whole_pass:
        from livetalking_amd.engine import Engine
        Engine.set_knob("PREFETCH", 0)
        try:
            conv_ms_whole = eng.time_convs(nf_pass, 10)[0]
        finally:
            Engine.set_knob("PREFETCH", 1)
    out = None
    if ranks.rank == 0:
        out = {
            "metric": "inferfps", "value": round(value, 2), "unit": "frames/s", "n_gpus": ranks.world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed_max / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"wav2lip256, {S} session(s)/GPU, {B}-frame batch, fp16 activations / fp32 accumulate, through LipReal.inference_batch",
                       "sessions_per_gpu": S, "batch": B, "frames_per_step_per_gpu": frames_per_step,
                       "parallelism": f"session-sharded x{ranks.world} (no collective)"},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_F16_TFLOPS, 5), "traffic": None,
                         "kernel": "conv3_kernel + conv7 / audio0 / audio3 / convs2d_kernel + rowgemm / rowconv_kernel (the 54 conv/convT layers + fused head = one pass)",
                         "pass_ms": round(conv_ms, 4),
                         # the stable cross-round figure: the pass with every call running the whole network (rounds 1-4 measured exactly this)
                         "conv_stack_ms": round(conv_ms_whole if conv_ms_whole else conv_ms, 4),
                         "pass_ms_note": "pass_ms: device time of one ltk_wav2lip_infer pass as the timed calls enqueue it (mel pack + conv launches + fused "
                                         "head, graph replay under knob GRAPH; with knob PREFETCH the steady state of a session's pipelined calls: `frac` is "
                                         "computed from it); conv_stack_ms: the same pass with every call running the whole network on its own (the figure of "
                                         "rounds 1-4; equals pass_ms when nothing was pipelined or under --no-whole-pass)",
                         # the roofline fraction recomputed from the TIMED LINE's own clock (ms_per_step: host path included) - the same measurement the
                         # driver's wall clock brackets; `frac` above is the device-pass figure
                         "frac_from_ms_per_step": round(2.0 * conv_macs / nf_pass * frames_per_step / (elapsed_max / args.steps) / 1e12 / PEAK_F16_TFLOPS, 5),
                         "frames_per_pass": nf_pass, "flops_per_frame": 2.0 * conv_macs / nf_pass,
                         "hipgraph": bool(graphs_timed_run), "graphs_captured_in_timed_run": graphs_timed_run,
                         "face_cache": False,
                         "pipelined_across_calls": bool(pf_stats["hits"] > 0),
                         "prefetch": dict(pf_stats, note="knob PREFETCH: while a call runs its audio encoder + decoder, the face encoder of the frames "
                                          "the session's NEXT call will ask for (bank frames index+B.., known from the bank walk) runs beside it on a third "
                                          "stream; hits = calls that started at the decoder.  Every layer runs once per frame and step inside the timed "
                                          "region (the first call runs the whole pass, the last call's prefetch is extra work); frames byte-identical to "
                                          "the knob off (tests/test_wav2lip_gpu.py::test_prefetched_face_encoder_equals_whole_pass)"),
                         "pass_ms_whole_pass_per_call": round(conv_ms_whole, 4) if conv_ms_whole else None},
            "per_rank_fps": [round(args.steps * frames_per_step / t, 1) for t in per_rank],
            "scheduler": sched,
        }
        if face_cache:
            # the session calls above skipped the face encoder (cached per bank frame): no roofline figure for this entry - the
            # HIP-event pass time below is the FULL pass on dummy inputs and `value` is not algorithmic FLOPs over time
            out["face_cache"] = {"on": True, "bytes_per_bank_frame": 4146176,
                                 "cache_bytes": int(eng.face_cache_bytes(sessions[0]._aid)) if hasattr(sessions[0], "_aid") else None,
                                 "full_pass_ms": round(conv_ms, 4)}
            out["roofline"] = None
            out["config"]["workload"] += ", LTK_FACE_CACHE=1 (deployment mode: face-encoder skip tensors cached per bank frame)"
        if sustained is not None:
            out["sustained"] = sustained
        if paced is not None:
            out["paced"] = paced
        if pcie is not None:
            out["pcie_inclusive"] = pcie
    for eng_ in model.engines:
        eng_.close()
    return out


def run_musetalk(args, ranks: Ranks, shared=None):
    """BASELINE.json configs[2] / configs[4]: MuseTalk (Whisper audio feat + U-Net + VAE decoder).  A step = one
    MuseReal.inference_batch per session (latent gather + PE + U-Net + VAE decode + uint8 BGR) with latents, weights and
    whisper chunks resident in HBM; the Whisper step (run_step's work, outside inferfps in the reference too) is timed
    beside it."""
    import argparse as ap
    import numpy as np
    import torch
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    dev = rank_device(ranks.local_rank)
    if dev >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {ranks.rank} needs GPU {dev}, this box has {torch.cuda.device_count()}")
    torch.cuda.set_device(dev)
    os.environ["LTK_DEVICE"] = str(dev)
    os.environ["LTK_MT_FP8"] = "1" if args.fp8 else "0"
    import livetalking_amd.avatars.musetalk_avatar as plugin
    import synth_inputs as synth

    S, B = args.sessions, args.batch
    fps_step = S * B
    shared = shared if shared is not None else {}
    if "unet" not in shared:
        shared["unet"], shared["vae"], shared["whisper"] = (synth.musetalk_unet_state_dict(), synth.vae_decoder_state_dict(),
                                                           synth.whisper_encoder_state_dict())
    model = plugin.load_model(shared["unet"], shared["vae"], shared["whisper"], max_frames=min(fps_step, 64), device=dev)
    eng = model.engine
    macs_all, macs_fp8 = eng.musetalk_info()
    n = 8
    lats = synth.musetalk_latents(n)
    frames, _, _ = synth.wav2lip_avatar(n_frames=n, full_hw=(720, 1280), box=320, seed=0)
    avatar = (frames, [np.full((480, 480, 3), 128, np.uint8)] * n, [(480, 200, 800, 520)] * n, [(400, 120, 880, 600)] * n, lats)
    sessions = []
    for s in range(S):
        sessions.append(plugin.MuseReal(ap.Namespace(fps=25, batch_size=B, l=10, r=10, sessionid=s), model, avatar))
    d_feat = torch.from_numpy(synth.musetalk_whisper_feats(fps_step)).cuda().reshape(S, B, 50, 384)
    drv = SessionThreads(sessions, [d_feat[s] for s in range(S)], stride=3)
    # untimed priming in front of the W warm-up steps: the MuseTalk pass is captured as a hipGraph the second time a frame count is
    # seen, and a capture inside the timed region would be timed
    PRIME = 3 if S == 1 else 8      # several session threads: their coalesced calls come in sizes 16 .. 16 S, each captured on its second sighting
    for i in range(PRIME):
        drv.step(i)
    for i in range(args.warmup):
        drv.step(PRIME + i)
    ranks.barrier()
    t0 = time.perf_counter()
    drv.step(PRIME + args.warmup, nsteps=args.steps)
    torch.cuda.synchronize()
    own = time.perf_counter() - t0
    ranks.barrier()
    elapsed = time.perf_counter() - t0
    elapsed_max, _ = ranks.gather_max(elapsed)
    _, per_rank = ranks.gather_max(own)
    value = ranks.world * args.steps * fps_step / elapsed_max
    sched = dict(sessions[0]._sched.stats)
    drv.close()
    nt = min(fps_step, 64)
    ms, macs = eng.musetalk_time(nt, 3)
    achieved = 2.0 * macs / (ms * 1e-3) / 1e12
    # roofline peak: MAC-weighted blend of the fp8 peak (layers on e4m3 operands) and the fp16 peak (everything else)
    f8 = macs_fp8 / macs_all if args.fp8 else 0.0
    peak = 1.0 / (f8 / PEAK_FP8_TFLOPS + (1.0 - f8) / PEAK_F16_TFLOPS)
    # WhisperASR.run_step's feature work (log-mel + whisper-tiny encoder + chunk slicing) for one session's 0.64-s step: the
    # reference runs it on the render thread, outside inferfps (base_avatar.py:364-373 times inference_batch only)
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
    out = None
    if ranks.rank == 0:
        out = {"metric": "inferfps", "value": round(value, 2), "unit": "frames/s", "n_gpus": ranks.world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(elapsed_max / args.steps * 1e3, 4), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "fp8(e4m3)+f16" if args.fp8 else "f16", "data": "synthetic",
               "config": {"workload": f"musetalk (U-Net + VAE decoder), {S} session(s)/GPU, {B}-frame batch, through MuseReal.inference_batch, " +
                                      (f"fp8 e4m3 operands on the resnet 3x3 convs ({f8:.0%} of the MACs), fp16 elsewhere, fp32 accumulate"
                                       if args.fp8 else "fp16 activations / fp32 accumulate"),
                          "sessions_per_gpu": S, "batch": B, "frames_per_step_per_gpu": fps_step,
                          "parallelism": f"session-sharded x{ranks.world} (no collective)"},
               "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                            "frac": round(achieved / peak, 5), "traffic": None,
                            "peak_note": "MAC-weighted blend of 5 PF (fp8 layers) and 2.5 PF (fp16 layers)" if args.fp8 else "dense fp16 MFMA",
                            "kernel": "conv3_kernel / conv_mfma_kernel (U-Net + VAE conv and linear layers; attention excluded from the flop count)",
                            "pass_ms": round(ms, 4), "frames_per_pass": nt, "flops_per_frame": 2.0 * macs / nt},
               "sessions_25fps": {"per_gpu": int(value / ranks.world // 25), "note": "saturating rate / 25 fps"},
               "per_rank_fps": [round(args.steps * fps_step / t, 1) for t in per_rank],
               "scheduler": sched,
               "whisper": {"ms_per_session_step": round(whisper_ms, 3), "frames_per_step": B,
                           "note": "log-mel + whisper-tiny encoder (1500 tokens) + chunk slicing, host PCM in, device features out; not in `value`",
                           "fps_incl_whisper": round(fps_step / (elapsed_max / args.steps + S * whisper_ms * 1e-3), 2)}}
    for e in model.engines:
        e.close()
    return out


def delivered_musetalk(args, shared, fp8, counts=(4, 16, 24, 32, 40)):
    """configs[4]'s deployment shape measured, not derived: S MuseTalk sessions on one GPU, one thread each, per 0.64-s period what
    the reference's render / inference / process threads do for a session (avatars/base_avatar.py:326-381, 383-467;
    avatars/audio_features/whisper.py:35-76): the Whisper feature step on the period's PCM (host PCM in, device chunks out),
    `MuseReal.inference_batch` for B frames, then `paste_back_frame` for each of them (resize + paste + blendLinear on the GPU, the
    batch's composites in one pinned device-to-host copy): B host 720p BGR frames per session and period.  A session count is
    sustained when every session has its frames before the next period starts in every measured period."""
    import argparse as ap
    import numpy as np
    import torch
    os.environ["LTK_MT_FP8"] = "1" if fp8 else "0"
    import livetalking_amd.avatars.musetalk_avatar as plugin
    from livetalking_amd.hostshim import mirror_index
    import synth_inputs as synth
    B = args.batch
    model = plugin.load_model(shared["unet"], shared["vae"], shared["whisper"], max_frames=64, device=0)
    eng = model.engine
    n = 8
    lats = synth.musetalk_latents(n)
    frames, _, _ = synth.wav2lip_avatar(n_frames=n, full_hw=(720, 1280), box=320, seed=0)
    avatar = (frames, [np.full((480, 480, 3), 128, np.uint8)] * n, [(480, 200, 800, 520)] * n, [(400, 120, 880, 600)] * n, lats)
    period = B / 25.0
    pcm = synth.synthetic_audio(2.0)[: (20 + 2 * B) * 320]
    sessions, results = [], []
    for S in counts:
        while len(sessions) < S:
            sessions.append(plugin.MuseReal(ap.Namespace(fps=25, batch_size=B, l=10, r=10, sessionid=len(sessions)), model, avatar))
        d_chunks = [torch.zeros(B, 50, 384, dtype=torch.float32, device="cuda") for _ in range(S)]
        periods = 4                                    # the first one is warm-up
        go = threading.Barrier(S + 1)
        lat = [[0.0] * periods for _ in range(S)]
        parts = [[0.0, 0.0, 0.0] for _ in range(S)]    # seconds in whisper / inference_batch / paste_back_frame (steady periods)
        errs = []
        t_start = [0.0]

        def work(i):
            sess = sessions[i]
            try:
                go.wait()
                for p in range(periods):
                    due = t_start[0] + p * period
                    while time.perf_counter() < due:
                        time.sleep(0.0005)
                    index = (p * B + 3 * i) % (2 * n)
                    t0 = time.perf_counter()
                    eng.whisper_step(pcm, B, first_row=10, d_out_ptr=d_chunks[i].data_ptr())
                    t1 = time.perf_counter()
                    pred = sess.inference_batch(index, d_chunks[i])
                    t2 = time.perf_counter()
                    chk = 0
                    for k in range(B):
                        frame = sess.paste_back_frame(pred[k], mirror_index(n, index + k))
                        chk += int(frame[0, 0, 0])
                    t3 = time.perf_counter()
                    lat[i][p] = t3 - due
                    if p:
                        parts[i][0] += t1 - t0; parts[i][1] += t2 - t1; parts[i][2] += t3 - t2
            except Exception as ex:  # noqa: BLE001
                errs.append(repr(ex))

        th = [threading.Thread(target=work, args=(i,), daemon=True) for i in range(S)]
        for t in th:
            t.start()
        t_start[0] = time.perf_counter() + 0.05
        go.wait()
        for t in th:
            t.join(timeout=120)
        if errs:
            results.append({"sessions": S, "error": errs[0][:200]})
            break
        worst = max(max(l[1:]) for l in lat)
        ok = worst < period
        k = (periods - 1) * S
        results.append({"sessions": S, "sustained": bool(ok), "latency_ms_max": round(worst * 1e3, 1),
                        "latency_ms_mean": round(float(np.mean([np.mean(l[1:]) for l in lat])) * 1e3, 1),
                        "finalfps_per_session": round(B / max(period, worst), 2),
                        "ms_per_session_period": {"whisper_step": round(sum(q[0] for q in parts) / k * 1e3, 1),
                                                  "inference_batch": round(sum(q[1] for q in parts) / k * 1e3, 1),
                                                  "paste_back_frames": round(sum(q[2] for q in parts) / k * 1e3, 1)}})
        if not ok:
            break
    for e in model.engines:
        e.close()
    best = max([r["sessions"] for r in results if r.get("sustained")], default=0)
    return {"max_sessions_25fps_delivered": best, "period_ms": period * 1e3, "fp8": bool(fp8), "tested": results,
            "largest_tested_count_sustained": bool(results and results[-1].get("sustained")),          # True: the list ended before the capacity did
            "note": "plugin level, measured: per session and 0.64-s period one Whisper feature step (host PCM in), one MuseReal.inference_batch "
                    "(16 frames) and 16 paste_back_frame composites (blend on the GPU, one pinned device-to-host copy per batch) = 16 host "
                    "720p BGR frames; one Python thread per session; period 0 excluded"}


def paced_capacity(args):
    """The largest number of 25-fps wav2lip256 sessions ONE GPU sustains: every session asks for its next B frames once
    per B/25 s, all requests of a period go down coalesced (<= 4096 frames per engine call); a session count is
    sustained when every period's last frame is ready before the next period starts.  Engine level (hundreds of Python
    session threads would measure the GIL, and those threads are the reference's own unchanged code)."""
    import torch
    import synth_inputs as synth
    from livetalking_amd.engine import Engine
    B = args.batch
    os.environ.setdefault("LTK_MICROBATCH", "256")
    eng = Engine(0)
    eng.load_wav2lip(synth.wav2lip_state_dict(1234), max_frames=4096)
    frames, faces, coords = synth.wav2lip_bank(n_frames=BANK_FRAMES, full_hw=(720, 1280), box=320, seed=0)
    aid = eng.register_avatar(faces, frames, coords)
    SMAX = 1024
    d_mel = torch.randn(B, 80, 16, dtype=torch.float32, device="cuda")
    d_pred = torch.zeros(4096 // B, B, 256, 256, 3, dtype=torch.uint8, device="cuda")      # outputs are overwritten per call
    period = B / 25.0
    per_call = 4096 // B

    def run_period(S, i):
        reqs = [(aid, i * B + 7 * s, B, d_mel.data_ptr(), d_pred[s % per_call].data_ptr()) for s in range(S)]
        for j in range(0, S, per_call):
            eng.wav2lip_infer(reqs[j:j + per_call])

    def sustained(S, periods=3):
        run_period(S, 0)
        t0 = time.perf_counter() + 0.02
        worst = 0.0
        for p in range(periods):
            due = t0 + p * period
            while time.perf_counter() < due:
                time.sleep(0.0005)
            run_period(S, 1 + p)
            worst = max(worst, time.perf_counter() - due)
        return worst < period, worst

    tested = []
    lo, hi = 16, SMAX            # invariant: lo sustained (or smallest), hi not (or the cap)
    ok, w = sustained(lo)
    tested.append((lo, round(w * 1e3, 1), ok))
    best_lat = w if ok else None
    if ok:
        while hi - lo > 16:
            mid = (lo + hi) // 2 // 16 * 16
            ok, w = sustained(mid)
            tested.append((mid, round(w * 1e3, 1), ok))
            if ok:
                lo, best_lat = mid, w
            else:
                hi = mid
    eng.close()
    return {"max_sessions_25fps": lo if best_lat is not None else 0, "period_ms": period * 1e3,
            "latency_ms_at_max": round(best_lat * 1e3, 1) if best_lat is not None else None,
            "tested": [{"sessions": s, "latency_ms": l, "sustained": o} for s, l, o in tested],
            "note": "bisection in steps of 16 sessions, 3 paced periods per trial, requests of a period coalesced (engine level)"}


def delivered_capacity(args):
    """DELIVERABLE session capacity of one GPU, plugin level: S sessions, each with its own thread doing per 0.64-s period
    what the reference's inference thread and process thread do for it (avatars/base_avatar.py:326-381, 383-467):
    `inference_batch` for B frames, then one host frame per prediction - `paste_back_frame` (the composited 720p BGR frame
    `output.push_video_frame` receives, server/webrtc.py:144-151) or, with the plugin's opt.egress = "i420", the device
    egress path (composite + watermark + BGR->I420 on the GPU, half the PCIe bytes).  A session count is sustained when every
    session has its B host frames before the next period starts (finalfps >= 25) in every measured period.  Both frame
    formats run in this one process (one model, one 250-frame bank); the pinned host pool is warmed before a count is timed
    (a deployment's sessions do not all start in the same 640 ms)."""
    import argparse as ap
    import numpy as np
    import torch
    os.environ.setdefault("LTK_MICROBATCH", "256")
    import livetalking_amd.avatars.wav2lip_avatar as plugin
    from livetalking_amd.hostshim import mirror_index
    import synth_inputs as synth
    B = args.batch
    # --pool N: the reference's deployment shape widened to N engines in ONE process (sharding.EnginePool: what app.py's single
    # process runs on an N-GPU node); on a 1-GPU box the N engines all live on GPU 0 (LTK_DEVICES=0,0,...): what is measured is then
    # the HOST side of the pool - N schedulers, N enqueue locks, one GIL - the GPU being shared
    pool = max(0, int(getattr(args, "pool", 0) or 0))
    if pool > 1:
        os.environ["LTK_DEVICES"] = ",".join(["0"] * pool)
        model = plugin.load_model(None, state_dict=synth.wav2lip_state_dict(1234), max_frames=256)
        assert len(model.engines) == pool
        eng = model.engines[0]
    else:
        model = plugin.load_model(None, state_dict=synth.wav2lip_state_dict(1234), max_frames=256, device=0)
        eng = model.engine
    plugin.warm_up(B, model, 256)
    avatar = synth.wav2lip_bank(n_frames=BANK_FRAMES, full_hw=(720, 1280), box=320, seed=0)
    H, W = avatar[0][0].shape[:2]
    audio = synth.synthetic_audio(4.0)
    starts = [int(16 + i * 3.2) for i in range(B)]
    d_mel = torch.zeros(16, B, 80, 16, dtype=torch.float32, device="cuda")
    for s in range(16):
        off = (s * 977) % (len(audio) - (20 + 2 * B) * 320)
        eng.mel_step(audio[off: off + (20 + 2 * B) * 320], starts, d_mel[s].data_ptr())
    sessions, egs = [], []
    period = B / 25.0

    def run_format(egress_fmt, counts):
        frame_bytes = H * W * 3 // 2 if egress_fmt == "i420" else H * W * 3
        results = []
        for S in counts:
            while len(sessions) < S:
                o = ap.Namespace(fps=25, batch_size=B, l=10, r=10, sessionid=len(sessions), egress="i420")
                sessions.append(plugin.LipReal(o, model, avatar))
                egs.append(None)
            if egress_fmt:
                for i in range(S):
                    if egs[i] is None:
                        egs[i] = sessions[i]._make_egress()      # opt.egress = "i420": the device-side process_frames path (egress.py)
            # warm the pinned host pool: S blocks of one batch each, allocated together, then returned to torch's caching allocator
            shape = (B, H * 3 // 2, W) if egress_fmt == "i420" else (B, H, W, 3)
            warm = [torch.empty(shape, dtype=torch.uint8, pin_memory=True) for _ in range(S)]
            del warm
            periods = 4                                            # the first one is warm-up (stream and scratch pools)
            go = threading.Barrier(S + 1)
            lat = [[0.0] * periods for _ in range(S)]
            infer_s = [0.0] * S
            errs = []
            t_start = [0.0]

            def work(i):
                sess = sessions[i]
                try:
                    go.wait()
                    for p in range(periods):
                        due = t_start[0] + p * period
                        while time.perf_counter() < due:
                            time.sleep(0.0005)
                        index = (p * B + 7 * i) % (2 * BANK_FRAMES)
                        t0 = time.perf_counter()
                        pred = sess.inference_batch(index, d_mel[i % 16])
                        infer_s[i] += time.perf_counter() - t0
                        chk = 0
                        for k in range(B):
                            if egress_fmt:
                                frame = egs[i].speaking_frame_of(pred[k], mirror_index(BANK_FRAMES, index + k))
                                chk += int(frame.reshape(-1)[0])
                            else:
                                frame = sess.paste_back_frame(pred[k], mirror_index(BANK_FRAMES, index + k))
                                chk += int(frame[0, 0, 0])        # the host array is real
                        lat[i][p] = time.perf_counter() - due
                except Exception as ex:  # noqa: BLE001
                    errs.append(repr(ex))

            th = [threading.Thread(target=work, args=(i,), daemon=True) for i in range(S)]
            for t in th:
                t.start()
            t_start[0] = time.perf_counter() + 0.05
            go.wait()
            for t in th:
                t.join(timeout=120)
            if errs:
                results.append({"sessions": S, "error": errs[0][:200]})
                break
            steady = [max(l[1:]) for l in lat]
            worst = max(steady)
            mean_lat = float(np.mean([np.mean(l[1:]) for l in lat]))
            ok = worst < period
            results.append({"sessions": S, "sustained": bool(ok), "latency_ms_max": round(worst * 1e3, 1), "latency_ms_mean": round(mean_lat * 1e3, 1),
                            "finalfps_per_session": round(B / max(period, worst), 2),
                            "inferfps_per_session_min": round(periods * B / max(infer_s), 1),
                            "first_period_ms_max": round(max(l[0] for l in lat) * 1e3, 1),
                            "d2h_GBps_needed": round(S * B * frame_bytes / period / 1e9, 2)})
            if not ok:
                break
        best = max([r["sessions"] for r in results if r.get("sustained")], default=0)
        return {"max_sessions_25fps_delivered": best, "frame_format": egress_fmt or "bgr24 (paste_back_frame)", "frame_bytes": frame_bytes,
                "largest_tested_count_sustained": bool(results and results[-1].get("sustained")),      # True: the list ended before the capacity did
                "tested": results}

    custom = [int(v) for v in args.delivered_sessions.split(",")] if args.delivered_sessions else None
    fmts = [f for f in args.delivered_formats.split(",") if f]
    out = {"period_ms": period * 1e3, "bank_frames": BANK_FRAMES, "pool_engines": max(1, pool),
           "face_cache": os.environ.get("LTK_FACE_CACHE", "0") not in ("", "0"),
           "bgr24": run_format("", custom or [384, 448, 512]) if "bgr24" in fmts else None,
           "i420": run_format("i420", custom or [384, 448, 512]) if "i420" in fmts else None,
           "note": "plugin level: per session and 0.64-s period one LipReal.inference_batch (16 frames) + 16 host frames (bgr24: paste_back_frame - "
                   "the batch's composites on the GPU, one pinned device-to-host copy; i420: the plugin's opt.egress path, + watermark + BGR->I420 "
                   "on the GPU); one Python thread per session; pinned pool warmed, period 0 excluded"}
    if pool > 1:       # per engine: sessions placed on it and what its scheduler coalesced
        per = []
        for k in range(pool):
            ss = [x for x in sessions if x._slot == k]
            st = dict(ss[0]._sched.stats) if ss else {}
            per.append({"engine": k, "sessions": len(ss), "calls": st.get("calls"), "requests": st.get("requests"), "frames": st.get("frames"),
                        "max_requests_per_call": st.get("max_requests_per_call")})
        out["per_engine"] = per
    for g in egs:
        if g is not None:
            g.close()
    for e in model.engines:
        e.close()
    return out


# ---------------------------------------------------------------------------------------------------------------
# CPU baseline, HBM traffic
# ---------------------------------------------------------------------------------------------------------------
def cpu_baseline(batch: int):
    """LipReal.inference_batch on the host cores: the reference's own class from a LiveTalking checkout when one is
    importable (kind "reference": build container), else the oracle restatement (kind "port": the GPU box).  fp32
    torch CPU, all cores, seeded synthetic weights / bank / audio; median of 3 after one warm-up, B=batch and B=1."""
    import numpy as np
    import torch
    import synth_inputs as synth
    from oracle import mel_oracle, plugin_oracle   # the checker, timed here as the CPU baseline only
    torch.set_num_threads(min(64, os.cpu_count() or 1))   # more threads only add contention on this model size
    sd_np = synth.wav2lip_state_dict(1234)
    sd = {k: torch.from_numpy(v) for k, v in sd_np.items()}
    frames, faces, coords = synth.wav2lip_avatar(n_frames=8, full_hw=(360, 640), box=160, seed=0)
    audio = synth.synthetic_audio(2.0)
    kind, call = "port", None
    # LTK_CPU_BASELINE_KIND=port times the oracle port even where a checkout exists (scripts/cpu_baseline_compare.py: both kinds in
    # the build container, so that the GPU box's `kind: port` figure is a validated proxy of the reference class)
    if os.environ.get("LTK_CPU_BASELINE_KIND", "") != "port" and os.path.exists(os.path.join(REF_ROOT, "avatars", "wav2lip_avatar.py")):
        try:
            from oracle import ref_loop
            cwd = os.getcwd()
            ref_loop.enter_reference(REF_ROOT)
            import avatars.wav2lip_avatar as ref_plugin
            from avatars.wav2lip.models import Wav2Lip
            os.chdir(cwd)
            net = Wav2Lip().eval()
            net.load_state_dict(sd)
            lip = ref_plugin.LipReal.__new__(ref_plugin.LipReal)
            lip.model = net
            lip.frame_list_cycle, lip.face_list_cycle, lip.coord_list_cycle = frames, faces, coords

            def call(index, B, feats):          # noqa: F811
                lip.batch_size = B
                return lip.inference_batch(index, feats)
            kind = "reference"
        except Exception:  # noqa: BLE001 - fall back to the port
            call = None
    if call is None:
        def call(index, B, feats):
            return plugin_oracle.inference_batch(sd, faces, index, B, feats)

    def median_fps(B):
        feats = mel_oracle.mel_chunks(audio[: (20 + 2 * B) * 320], 20 + 2 * B)
        call(0, B, feats)
        ts = []
        for i in range(3):
            t0 = time.perf_counter()
            call((i + 1) * B, B, feats)
            ts.append(time.perf_counter() - t0)
        return B / float(np.median(ts))

    fb, f1 = median_fps(batch), median_fps(1)
    what = "the reference's LipReal.inference_batch (avatars/wav2lip_avatar.py:116-139)" if kind == "reference" else \
        "oracle port of LipReal.inference_batch"
    return {"value": round(fb, 3), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": kind,
            "sample": f"median of 3 x {what}, B={batch}, fp32 torch-CPU, seeded synthetic weights/bank/audio",
            "b1": {"value": round(f1, 3), "unit": "frames/s", "sample": "same, B=1 (BASELINE.json configs[0])"}}


def is_layer_kernel(name):
    """A kernel that runs one of the 54 conv / convT layers of a Wav2Lip pass (or its fused head): conv7 / conv3 / convs2d / conv_mfma /
    rowconv + rowgemm / audio0 + audio3 - templated kernels print demangled ("conv3_kernel<...>"), the others mangled ("_ZN3ltk14convs2d_kernel...")."""
    return "conv" in name or "rowgemm" in name or "audio0_kernel" in name or "audio3_kernel" in name


def measure_traffic(sub, extra, passes, conv_only):
    """HBM bytes per pass from rocprofv3 PMC counters, collected as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in
    SEPARATE --pmc passes (TCC slots), FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B), KiB units.  Each pass
    profiles `bench.py --sub <sub>` (`passes` timed passes + 1 warm pass of the same workload and nothing else).
    conv_only: count the conv kernels (Wav2Lip conv stack); else every kernel of the run except the runtime's copy / fill
    blits of the model load."""
    import glob
    import shutil
    import sqlite3
    import tempfile
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not found"
    tot, heads = {}, {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="ltk_pmc_")
        cmd = [exe, "--pmc", counter, "-d", d, "-o", "r", "--", sys.executable, os.path.abspath(__file__), "--sub", sub,
               "--steps", str(passes)] + extra
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
        except subprocess.TimeoutExpired:
            return None, f"rocprofv3 --pmc {counter} timed out"
        dbs = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)
        if r.returncode != 0 or not dbs:
            return None, f"rocprofv3 --pmc {counter} failed (rc {r.returncode})"
        db = sqlite3.connect(dbs[0])
        names = [t[0] for t in db.execute("select name from sqlite_master where type in ('table','view')")]
        if "counters_collection" not in names:
            return None, "no counters_collection view in the rocprofv3 database"
        v = 0.0
        for k, c, val, cnt in db.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection group by kernel_name, counter_name"):
            if c != counter:
                continue
            if "conv3_head_kernel" in k:
                heads[counter] = heads.get(counter, 0) + int(cnt)       # one fused-head launch per Wav2Lip pass: the pass count of the run
            if conv_only and not is_layer_kernel(k):
                continue
            if not conv_only and ("__amd_rocclr" in k or "debug" in k):
                continue
            v += float(val)
        tot[counter] = v
        shutil.rmtree(d, ignore_errors=True)
    # passes of the profiled body: counted from the trace for Wav2Lip (one fused-head dispatch per pass; the body runs whole passes -
    # knob PREFETCH off - so every pass is the 54 layers once: `passes` + 2 warm ones), `passes` + the two warm runs of
    # ltk_musetalk_time for MuseTalk
    n = passes + 2
    if sub == "convpasses" and heads.get("FETCH_SIZE"):
        n = heads["FETCH_SIZE"]
    rd = tot["FETCH_SIZE"] * 1024.0 * 2.0 / n
    wr = tot["WRITE_SIZE"] * 1024.0 / n
    return rd + wr, {"read_bytes": rd, "write_bytes": wr, "passes_profiled": n,
                     "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), "
                               + ("conv kernels only" if conv_only else "all kernels of the pass") + ", FETCH_SIZE x2 (gfx950), KiB units"}


def sub_convpasses(args):
    """Body profiled by measure_traffic: K passes of the conv stack, nothing else."""
    import synth_inputs as synth
    from livetalking_amd.engine import Engine
    nf = min(args.sessions * args.batch, 256)
    eng = Engine(0)
    Engine.set_knob("GRAPH", 0)             # counters per kernel dispatch: the same launches, issued one by one
    Engine.set_knob("PREFETCH", 0)          # whole passes: N passes = N runs of every layer (pipelined passes carry the NEXT pass's face encoder)
    eng.load_wav2lip(synth.wav2lip_state_dict(1234), max_frames=nf)
    eng.time_convs(nf, args.steps)          # = 2 warm passes + `steps` passes
    eng.close()


def sub_mtpasses(args):
    """Body profiled by measure_traffic: K MuseTalk passes (U-Net + VAE decoder) of min(sessions x batch, 64) frames."""
    import synth_inputs as synth
    from livetalking_amd.engine import Engine
    nt = min(args.sessions * args.batch, 64)
    eng = Engine(0)
    Engine.set_knob("GRAPH", 0)             # counters per kernel dispatch: the same launches, issued one by one
    eng.load_musetalk(synth.musetalk_unet_state_dict(), synth.vae_decoder_state_dict(), max_frames=nt, fp8=args.fp8)
    eng.musetalk_time(nt, args.steps)       # = 2 warm passes + `steps` passes
    eng.close()


def run_sub(name, extra, timeout=600, env=None):
    cmd = [sys.executable, os.path.abspath(__file__), "--sub", name] + extra
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=dict(os.environ, **env) if env else None)
    except subprocess.TimeoutExpired:
        return {"error": f"{name}: timed out after {timeout} s"}
    for line in reversed(r.stdout.strip().splitlines()):
        if line.startswith("{") or line.startswith("["):
            try:
                return json.loads(line)
            except Exception:  # noqa: BLE001
                pass
    return {"error": f"{name}: rc {r.returncode}: {(r.stderr or r.stdout)[-400:]}"}


def dry_rank(args, ranks: Ranks):
    """The launcher / barrier / max-over-ranks protocol without a GPU: rank r 'works' 10 ms x (r + 1) per step."""
    ranks.barrier(sync_gpu=False)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.01 * (ranks.rank + 1))
    own = time.perf_counter() - t0
    ranks.barrier(sync_gpu=False)
    elapsed = time.perf_counter() - t0
    mx, _ = ranks.gather_max(elapsed)
    _, per_rank = ranks.gather_max(own)
    if ranks.rank == 0:
        return {"metric": "dry-ranks", "n_gpus": ranks.world, "steps": args.steps, "ms_per_step": round(mx / args.steps * 1e3, 3),
                "per_rank_ms_per_step": [round(t / args.steps * 1e3, 3) for t in per_rank], "ranks_seen": len(per_rank)}
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--sessions", type=int, default=1, help="sessions per GPU, each with its own inference thread")
    ap.add_argument("--batch", type=int, default=16, help="frames per session per step (opt.batch_size)")
    ap.add_argument("--model", choices=("wav2lip", "musetalk"), default="wav2lip",
                    help="wav2lip = BASELINE.json configs[1] (default, the driver's line); musetalk = configs[2]")
    ap.add_argument("--fp8", action="store_true", help="musetalk: BASELINE configs[4] fp8 conv path")
    ap.add_argument("--paced", type=int, default=0, help="after the timed run: N periods of B/25 s with every session paced at 25 fps")
    ap.add_argument("--paced-stagger", action="store_true", help="--paced with the sessions' request times spread evenly over the period (every request a call of its own)")
    ap.add_argument("--sustain", type=float, default=None,
                    help="seconds of the `sustained` twin behind the timed region (default 2.0 for the primary wav2lip line, 0 = off)")
    ap.add_argument("--no-whole-pass", action="store_true", help="skip the comparison timing with knob PREFETCH off (profiler runs: the last pass of "
                    "the process is then a pipelined one)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the other BASELINE configs (also[]) and the paced capacity")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 PMC passes (roofline.traffic = null)")
    ap.add_argument("--dry-ranks", action="store_true", help="launcher / barrier protocol only, no GPU (CPU test)")
    ap.add_argument("--delivered-sessions", default="", help="session counts of the delivered-capacity run (default: 384,448,512 for both frame formats)")
    ap.add_argument("--delivered-formats", default="bgr24,i420", help="frame formats of the delivered-capacity run")
    ap.add_argument("--pool", type=int, default=0, help="--sub delivered-capacity: N engines in this one process (EnginePool; on a 1-GPU box all on GPU 0)")
    ap.add_argument("--sub", default="", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.sustain is None:
        args.sustain = 0.0 if (args.sub or args.dry_ranks) else 2.0

    if args.sub == "convpasses":
        return sub_convpasses(args)
    if args.sub == "mtpasses":
        return sub_mtpasses(args)
    if args.sub == "paced-capacity":
        print(json.dumps(paced_capacity(args)), flush=True)
        return
    if args.sub == "delivered-capacity":
        print(json.dumps(delivered_capacity(args)), flush=True)
        return
    if args.sub == "musetalk-both":        # configs[2] then configs[4]'s share in one process: the synthetic weights are made once
        ranks = Ranks()
        shared = {}
        a2 = argparse.Namespace(**vars(args)); a2.sessions, a2.fp8, a2.steps, a2.warmup = 1, False, 4, 2
        o2 = run_musetalk(a2, ranks, shared)
        a3 = argparse.Namespace(**vars(args)); a3.sessions, a3.fp8, a3.steps, a3.warmup = 4, False, 3, 1
        o3 = run_musetalk(a3, ranks, shared)
        a4 = argparse.Namespace(**vars(args)); a4.sessions, a4.fp8, a4.steps, a4.warmup = 4, True, 3, 1
        o4 = run_musetalk(a4, ranks, shared)
        if isinstance(o4, dict):       # configs[4] as deployed: paced sessions with the Whisper step, the blend composite and the copy to the host
            try:
                o4["delivered"] = delivered_musetalk(a4, shared, fp8=True)
                o4["sessions_25fps"] = {"per_gpu_delivered": o4["delivered"]["max_sessions_25fps_delivered"],
                                        "per_gpu_saturating_rate_over_25": int(o4["value"] // 25),
                                        "note": "per_gpu_delivered: measured paced run (Whisper step + inference_batch + paste_back_frame + D2H per "
                                                "period); the other figure is the saturating inferfps / 25 (frames left on the device)"}
            except Exception as ex:  # noqa: BLE001
                o4["delivered"] = {"error": repr(ex)[:300]}
        print(json.dumps([o2, o3, o4]), flush=True)
        return

    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args, sys.argv[1:]))
    ranks = Ranks()
    if ranks.world != args.gpus and ranks.rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={ranks.world}: using WORLD_SIZE", file=sys.stderr)
    if args.dry_ranks:
        out = dry_rank(args, ranks)
    elif args.model == "musetalk":
        out = run_musetalk(args, ranks)
    else:
        out = run_wav2lip(args, ranks)
    ranks.close()
    if ranks.rank != 0 or out is None:
        return
    primary = not args.sub and not args.dry_ranks
    if primary and ranks.world == 1 and args.model == "wav2lip":
        def add_traffic(rl, sub, extra, passes, conv_only, what):
            traffic, info = measure_traffic(sub, extra, passes, conv_only)
            rl["traffic"] = traffic
            rl["traffic_unit"] = what
            rl["traffic_info"] = info

        if not args.no_traffic:
            add_traffic(out["roofline"], "convpasses", ["--sessions", str(args.sessions), "--batch", str(args.batch)], 6, True,
                        "HBM bytes per conv-stack pass")
        if not args.no_also:
            mt = run_sub("musetalk-both", ["--batch", str(args.batch)])
            also = [run_sub("also-w2l16", ["--sessions", "16", "--batch", str(args.batch), "--steps", "10", "--warmup", "12", "--paced", "4"])]
            also += mt if isinstance(mt, list) else [mt]
            tags = ["configs[3] per-GPU share: 16 wav2lip256 sessions on one GPU (16 session threads, continuous batching)",
                    "configs[2]: MuseTalk, 1 session",
                    "configs[4] per-GPU share WITHOUT fp8: 4 MuseTalk sessions, fp16 (the like-for-like partner of the next entry)",
                    "configs[4] per-GPU share: 4 MuseTalk sessions, fp8 conv path"]
            for a, t in zip(also, tags):
                if isinstance(a, dict):
                    a["baseline_config"] = t
            if not args.no_traffic:          # HBM bytes of the other BASELINE shares (each two short rocprofv3 --pmc runs)
                subs = [("convpasses", ["--sessions", "16", "--batch", str(args.batch)], 3, True, "HBM bytes per 256-frame conv-stack pass"),
                        ("mtpasses", ["--sessions", "1", "--batch", str(args.batch)], 2, False, "HBM bytes per 16-frame MuseTalk pass"),
                        None,
                        None]      # fp8, 4 sessions: profiles/r03_mtfp8_pmc.json (148 GB per 64-frame pass); two more 25-s profiler runs here would not fit "minutes"
                for a, sb in zip(also, subs):
                    if sb is not None and isinstance(a, dict) and isinstance(a.get("roofline"), dict):
                        add_traffic(a["roofline"], *sb)
            fc = run_sub("also-w2l16-facecache", ["--sessions", "16", "--batch", str(args.batch), "--steps", "10", "--warmup", "12", "--paced", "4"],
                         env={"LTK_FACE_CACHE": "1"})
            if isinstance(fc, dict):
                fc["baseline_config"] = ("configs[3] per-GPU share in the opt-in deployment mode LTK_FACE_CACHE=1 (not a benchmark line: the face "
                                         "encoder's outputs are cached per bank frame, 4.15 MB each; compare with also[0])")
            also.append(fc)
            out["also"] = also
            out["paced"] = run_sub("paced-capacity", ["--batch", str(args.batch)])
            out["delivered"] = run_sub("delivered-capacity", ["--batch", str(args.batch)])
            out["delivered_face_cache"] = run_sub("delivered-capacity", ["--batch", str(args.batch), "--delivered-sessions", "512",
                                                                         "--delivered-formats", "bgr24"], env={"LTK_FACE_CACHE": "1"})
            w16 = also[0] if isinstance(also[0], dict) else {}
            out["sessions_25fps"] = {"per_gpu_delivered": (out["delivered"].get("bgr24") or {}).get("max_sessions_25fps_delivered"),
                                     "per_gpu_delivered_i420": (out["delivered"].get("i420") or {}).get("max_sessions_25fps_delivered"),
                                     "per_gpu_kernel_capacity": out["paced"].get("max_sessions_25fps"),
                                     "at_16_sessions_per_gpu": w16.get("paced"),
                                     "note": "per_gpu_delivered: plugin level, every session gets its 16 composited 720p frames on the host per "
                                             "period (inference_batch + paste_back_frame, one thread per session); per_gpu_delivered_i420: the same "
                                             "through the plugin's opt.egress = i420 path (watermarked I420 frames, half the PCIe bytes); per_gpu_kernel_capacity: "
                                             "engine level, frames stay on the device (paced bisection); at_16_sessions_per_gpu: 16 paced "
                                             "session threads through LipReal.inference_batch"}
    if primary and ranks.world == 1 and not args.no_cpu_baseline:         # the CPU baseline is timed at N=1 only
        out["cpu_baseline"] = cpu_baseline(args.batch) if args.model == "wav2lip" else cpu_baseline_musetalk()
    print(json.dumps(out), flush=True)


def cpu_baseline_musetalk(budget_s: float = 25.0):
    """The oracle restatement of MuseReal.inference_batch (fp32, torch CPU) on the host cores, bounded sample."""
    import torch
    import synth_inputs as synth
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


if __name__ == "__main__":
    main()
