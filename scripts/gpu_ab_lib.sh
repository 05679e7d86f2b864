#!/bin/bash
# A/B of several builds of libltk_hip.so in one job (LTK_LIB selects the library), two interleaved rounds.
# usage: scripts/gpu_ab_lib.sh <tag> <a.so> <b.so> [...]      MT=0 skips the MuseTalk pass
TAG=$1; shift
mkdir -p gpurun_out
OUT=gpurun_out/${TAG}.log
: > $OUT
for rnd in 1 2; do
  for lib in "$@"; do
    echo "######## round $rnd lib $lib" >> $OUT
    LTK_LIB=$PWD/$lib ROUNDS=3 timeout 300 python scripts/layer_times.py "TILE_RULE=1" -- 16 256 2>&1 | grep -E "^(====|sum|conv stack|face_decoder_blocks.7|face_decoder_blocks.5.1|face_decoder_blocks.3.1|face_encoder_blocks.2.1|output)" >> $OUT
    if [ "${MT:-1}" != "0" ]; then
      LTK_LIB=$PWD/$lib timeout 300 python scripts/mt_op_times.py 16 2>&1 | grep -E "pass|conv/linear|GroupNorm|resnets.1.conv2|resnets.1.conv1" | head -8 >> $OUT
    fi
  done
done
cat $OUT
