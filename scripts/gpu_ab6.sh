cd /root/repo
timeout 1200 python -m pytest tests/test_musetalk_gpu.py tests/test_fp8_gpu.py tests/test_musetalk_plugin_gpu.py -m gpu -q -x 2>&1 | tail -3
timeout 300 python bench.py --model musetalk --steps 4 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mt fused', d['value'], d['ms_per_step'], d['roofline']['frac'])"
LTK_MT_NO_QKV_FUSE=1 timeout 300 python bench.py --model musetalk --steps 4 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mt unfused', d['value'], d['ms_per_step'], d['roofline']['frac'])"
timeout 300 python bench.py --model musetalk --fp8 --steps 4 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mt fp8', d['value'], d['ms_per_step'], d['roofline']['frac'])"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/mt_trace2 -o r -- python /root/repo/bench.py --model musetalk --steps 2 --warmup 1 > /root/repo/gpurun_out/mt_trace2.log 2>&1
