#!/bin/bash
# Kernel trace of one MuseTalk workload only (no PMC passes): `gpurun -- 'bash scripts/trace_mt.sh <tag> [bench args]'` writes the per-kernel
# stats and the launch timeline of the last pass to gpurun_out/<tag>_summary/ (scripts/make_profile_summary.py).  Environment knobs pass through.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-mtq}; shift
ARGS=${*:---model musetalk --steps 2 --warmup 1}
O=$R/gpurun_out; P=$O/$TAG; mkdir -p $P; cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $P/mt_trace -o r -- python $R/bench.py $ARGS --no-cpu-baseline --no-also --no-traffic > $P/mt_trace.log 2>&1
tail -2 $P/mt_trace.log | cut -c1-400
cd $R; mkdir -p $O/${TAG}_summary
python scripts/make_profile_summary.py $P $O/${TAG}_summary/$TAG --name mt --all-kernels --frames ${FRAMES:-16} --cmd "bench.py $ARGS" > /dev/null 2>$O/${TAG}_summary/err.txt; tail -3 $O/${TAG}_summary/err.txt
rm -rf $P; ls $O/${TAG}_summary
