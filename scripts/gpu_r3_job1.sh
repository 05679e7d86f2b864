#!/bin/bash
# round 3, job 1: baseline of this round's box + ablation of the small-map / tail layers
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_wav2lip_gpu.py -m gpu -q -x -k "one_and_two or reference_golden or fused_head" 2>&1 | tail -5 > $O/j1_pytest.log
timeout 400 python bench.py > $O/j1_bench.json 2> $O/j1_bench.err; tail -c 600 $O/j1_bench.json
ROUNDS=3 timeout 300 python scripts/layer_times.py "TILE_RULE=1" -- 16 > $O/j1_layer_times.txt 2>&1
LTK_LIB=$R/ab_libs/libltk_hip_ablate.so SWEEP_FRAMES=16 ABLATE_MASKS=0,1,2,3,4,7,16,64,68 timeout 600 python scripts/conv_ablate.py > $O/j1_ablate16.txt 2>&1
cat $O/j1_pytest.log; cat $O/j1_ablate16.txt
