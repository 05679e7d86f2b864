cd /root/repo
timeout 1200 python -m pytest tests/test_conv_gpu.py tests/test_wav2lip_gpu.py tests/test_fp8_gpu.py tests/test_musetalk_gpu.py -m gpu -q -x 2>&1 | tail -5
for i in 1 2; do
  timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('w2l s1', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
timeout 300 python bench.py --steps 10 --warmup 3 --sessions 16 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('w2l s16', d['value'], d['ms_per_step'], d['roofline']['frac'])"
for m in "" "--fp8"; do
  timeout 300 python bench.py --model musetalk --steps 4 --warmup 2 $m 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mt', '$m', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
timeout 300 python scripts/mt_layer_sweep.py 2>/dev/null | grep -v amdgpu
