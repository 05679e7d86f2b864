#!/usr/bin/env python3
"""Python time of one LipReal.inference_batch call around a no-op engine (CPU; no GPU, no library): what a single session adds between
two engine calls, i.e. GPU idle time on the 1-session bench line (ms_per_step - device pass).  With cProfile's top entries.

    LTK_ALLOW_STANDIN=1 python scripts/host_overhead.py
"""
import sys, time, types, cProfile, pstats
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from livetalking_amd.avatars import wav2lip_avatar as wa
from livetalking_amd.scheduler import get_scheduler

class NopEngine:
    device = 0; max_frames = 256
    torch_device = torch.device("cpu")
    def wav2lip_infer(self, reqs, stream=0): pass
    def paste_back_batch(self, *a, **k): pass
    def close(self): pass
eng = NopEngine()
s = wa.LipReal.__new__(wa.LipReal)
s.batch_size = 16; s.engine = eng; s._aid = 1; s._sched = get_scheduler(eng)
s.frame_list_cycle = [None] * 250
mel = torch.zeros(16, 80, 16)
for _ in range(2000): s.inference_batch(0, mel)
N = 20000
t0 = time.perf_counter()
for i in range(N): s.inference_batch(i * 16, mel)
t1 = time.perf_counter()
print("us per inference_batch (no-op engine):", (t1 - t0) / N * 1e6)
pr = cProfile.Profile(); pr.enable()
for i in range(5000): s.inference_batch(i * 16, mel)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
