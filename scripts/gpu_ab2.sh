# stride-2 selection rule A/B (same box): default (conv3 s2 only for Cin>=256) vs LTK_CONV_V3_S2=2 (always)
cd /root/repo
timeout 600 python -m pytest tests/test_conv_gpu.py tests/test_wav2lip_gpu.py -m gpu -q 2>&1 | tail -2
for mode in 1 2 1 2; do
  LTK_CONV_V3_S2=$mode timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('s2mode', $mode, d['value'], d['ms_per_step'])"
done
for mode in 1 2; do
  LTK_CONV_V3_S2=$mode timeout 300 python bench.py --steps 10 --warmup 3 --sessions 16 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('s2mode s16', $mode, d['value'], d['ms_per_step'])"
done
