#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_wav2lip_gpu.py tests/test_mel_paste_gpu.py tests/test_plugin_gpu.py -m gpu -q -s > $O/pytest_all.log 2>&1; echo "pytest exit $?" >> $O/pytest_all.log
grep -E "FAIL|EXC|passed|failed|rror|batching|\[infer\]" $O/pytest_all.log | head -40
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-330
timeout 300 python bench.py --steps 10 --warmup 3 --sessions 16 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-330
if [ -n "$MT" ]; then timeout 600 python -m pytest tests/test_musetalk_gpu.py -m gpu -q -s 2>&1 | grep -E "U-Net output|frames PSNR|passed|failed"; timeout 300 python bench.py --model musetalk --steps 5 --warmup 2 2>/dev/null | tail -1 | cut -c1-330; fi
