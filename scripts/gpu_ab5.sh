cd /root/repo
timeout 1200 python -m pytest tests/test_conv_gpu.py tests/test_musetalk_gpu.py tests/test_whisper_gpu.py -m gpu -q -x 2>&1 | tail -3
for v in 4 2; do
  LTK_CONV3_NBT=$v timeout 300 python bench.py --model musetalk --steps 4 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mt nbt$v', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
LTK_GEMM_NC8=8 timeout 300 python bench.py --model musetalk --steps 4 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mt nc8=8', d['value'], d['ms_per_step'], d['roofline']['frac'])"
timeout 300 python scripts/mt_layer_sweep.py 2>/dev/null | grep -v "amdgpu\|3x3"
