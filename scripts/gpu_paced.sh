cd /root/repo
for S in 560 608 640; do
  timeout 300 python bench.py --steps 3 --warmup 2 --sessions $S --paced 6 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('paced', d['config']['sessions_per_gpu'], d['value'], d['paced'])"
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/mt_trace3 -o r -- python /root/repo/bench.py --model musetalk --steps 2 --warmup 1 > /root/repo/gpurun_out/mt_trace3.log 2>&1
cd /root/repo
timeout 300 python bench.py --model musetalk --steps 6 --warmup 2 > gpurun_out/bench_mt_final.json 2>/dev/null
timeout 300 python bench.py --model musetalk --fp8 --steps 6 --warmup 2 > gpurun_out/bench_mt_fp8_final.json 2>/dev/null
