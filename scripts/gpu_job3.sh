#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -s > $O/pytest5.log 2>&1; echo "pytest exit $?" >> $O/pytest3.log
grep -E "FAIL|EXC|passed|failed|rror|batching" $O/pytest5.log | head -40
SWEEP_FRAMES=16 timeout 300 python scripts/conv_sweep.py > $O/sweep5.log 2>&1; cat $O/sweep5.log; timeout 300 python scripts/conv_ablate.py > $O/ablate5.log 2>&1; cat $O/ablate3.log
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/bench5_s1.json 2> $O/bench5_s1.err; cat $O/bench5_s1.json; tail -2 $O/bench5_s1.err
timeout 300 python bench.py --steps 10 --warmup 3 --sessions 16 --no-cpu-baseline > $O/bench5_s16.json 2> $O/bench5_s16.err; cat $O/bench5_s16.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof5_trace -o r -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/prof5_trace.log 2>&1
