#!/bin/bash
# 16 session threads through LipReal.inference_batch: saturating rate by scheduler window (in-job A/B, two rounds)
for rnd in 1 2; do for us in 0 200 500; do echo "== LTK_COALESCE_AUTO_US=$us"; LTK_COALESCE_AUTO_US=$us timeout 200 python bench.py --sessions 16 --steps 20 --warmup 5 --no-also --no-traffic --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['scheduler'], d['roofline']['conv_stack_ms'])"; done; done
