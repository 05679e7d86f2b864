cd /root/repo
timeout 1200 python -m pytest tests/test_musetalk_gpu.py tests/test_fp8_gpu.py tests/test_musetalk_plugin_gpu.py tests/test_whisper_gpu.py -m gpu -q 2>&1 | tail -3
timeout 300 python bench.py --model musetalk --steps 4 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mt', d['value'], d['ms_per_step'], d['roofline']['frac'])"
timeout 300 python bench.py --model musetalk --fp8 --steps 4 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mt fp8', d['value'], d['ms_per_step'], d['roofline']['frac'])"
