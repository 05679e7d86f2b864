#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out; mkdir -p $O
cd $R
for mb in 16 8 4; do
  echo "== microbatch $mb, 1 session"; LTK_MICROBATCH=$mb timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['conv_stack_ms'])"
done
for mb in 256 64 32 16 8; do
  echo "== microbatch $mb, 16 sessions"; LTK_MICROBATCH=$mb timeout 300 python bench.py --steps 8 --warmup 2 --sessions 16 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['conv_stack_ms'])"
done
