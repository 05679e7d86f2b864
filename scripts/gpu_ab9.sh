cd /root/repo
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_wav2lip_gpu.py tests/test_mel_paste_gpu.py -m gpu -q 2>&1 | tail -2
for lib in old new old new; do
  L=""; [ $lib = old ] && L=/root/repo/ab_libs/libltk_old.so
  LTK_LIB=$L timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('w2l s1 $lib', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
for lib in old new; do
  L=""; [ $lib = old ] && L=/root/repo/ab_libs/libltk_old.so
  LTK_LIB=$L timeout 300 python bench.py --steps 10 --warmup 3 --sessions 16 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('w2l s16 $lib', d['value'], d['ms_per_step'], d['roofline']['frac'])"
  LTK_LIB=$L timeout 300 python bench.py --model musetalk --steps 4 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mt $lib', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
