#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -s > $O/pytest9.log 2>&1; echo "pytest exit $?" >> $O/pytest9.log
grep -E "FAIL|EXC|passed|failed|rror|batching" $O/pytest9.log | head -40

timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/bench9_s1.json 2> $O/bench9_s1.err; cat $O/bench9_s1.json; tail -2 $O/bench9_s1.err
timeout 300 python bench.py --steps 10 --warmup 3 --sessions 16 --no-cpu-baseline > $O/bench9_s16.json 2> $O/bench9_s16.err; cat $O/bench9_s16.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof9_trace -o r -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/prof9_trace.log 2>&1
