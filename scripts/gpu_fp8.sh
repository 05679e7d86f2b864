cd /root/repo
timeout 900 python -m pytest tests/test_fp8_gpu.py -m gpu -x -q -s 2>&1 | grep -E "fp8|passed|failed|Error|error|assert" | head -60
for m in "" "--fp8" "" "--fp8"; do
  timeout 300 python bench.py --model musetalk --steps 4 --warmup 2 $m 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mt', '$m', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
