set -x
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -s > gpurun_out/pytest2.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest2.log
tail -4 gpurun_out/pytest2.log
python bench.py --steps 30 --warmup 5 > gpurun_out/bench_s1.log 2>&1; tail -2 gpurun_out/bench_s1.log
python bench.py --steps 10 --warmup 3 --sessions 16 --no-cpu-baseline > gpurun_out/bench_s16.log 2>&1; tail -1 gpurun_out/bench_s16.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_s1 -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_s1.log 2>&1
ls -R $GRAFT_REPO_ROOT/gpurun_out/prof_s1 | head -20
