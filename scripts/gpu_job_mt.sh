#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_musetalk_gpu.py -k vae_encoder -m gpu -q -s -x > $O/pytest_mt.log 2>&1; echo "pytest exit $?" >> $O/pytest_mt.log
grep -E "\[mt\]|whisper|FAIL|passed|failed|rror|assert" $O/pytest_mt.log | head -150
