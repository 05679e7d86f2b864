mkdir -p gpurun_out
python -m pytest tests -m gpu -q -s > gpurun_out/pytest4.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest4.log
tail -3 gpurun_out/pytest4.log
python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench4_s1.log 2>&1; tail -1 gpurun_out/bench4_s1.log
python bench.py --steps 10 --warmup 3 --sessions 16 --no-cpu-baseline > gpurun_out/bench4_s16.log 2>&1; tail -1 gpurun_out/bench4_s16.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof4 -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof4.log 2>&1
