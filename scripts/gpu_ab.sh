python -m pytest tests/test_conv_gpu.py tests/test_musetalk_gpu.py -m gpu -q 2>&1 | tail -2
for i in 1 2; do python bench.py --model musetalk --steps 4 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mt', d['value'], d['ms_per_step'])"; done
python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-60
