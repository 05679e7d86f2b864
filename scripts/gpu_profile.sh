#!/bin/bash
# Round profile: bench lines + rocprofv3 kernel trace + PMC passes (separate runs, as the pool requires)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
TAG=${1:-r02}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 900 python bench.py --steps 50 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; head -c 400 $O/bench_default.json; echo
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/mt_trace -o r -- python $R/bench.py --model musetalk --steps 2 --warmup 1 --no-cpu-baseline > $O/mt_trace.log 2>&1
BCMD="python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-also --no-traffic"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o r -- $BCMD > $O/trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o r -- $BCMD > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o r -- $BCMD > $O/pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/pmc_sq -o r -- $BCMD > $O/pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum -d $O/pmc_l2 -o r -- $BCMD > $O/pmc_l2.log 2>&1
ls $O
