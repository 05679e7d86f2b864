cd /root/repo
for t in 448 1100 2200 448 1100 2200; do
  LTK_CONV_PXW4_MIN=$t timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('w2l s1 pxw4min=$t', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
for t in 448 2200; do
  LTK_CONV_PXW4_MIN=$t timeout 300 python bench.py --steps 8 --warmup 3 --sessions 16 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('w2l s16 pxw4min=$t', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
