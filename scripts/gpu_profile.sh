#!/bin/bash
# Round profile: bench lines + rocprofv3 kernel trace + PMC passes (separate runs, as the pool requires)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
TAG=${1:-r01}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 600 python bench.py --steps 50 --warmup 5 > $O/bench_s1.json 2> $O/bench_s1.err; cat $O/bench_s1.json
timeout 300 python bench.py --steps 10 --warmup 3 --sessions 16 --no-cpu-baseline > $O/bench_s16.json 2> $O/bench_s16.err; cat $O/bench_s16.json
timeout 300 python bench.py --model musetalk --steps 6 --warmup 2 > $O/bench_mt.json 2> $O/bench_mt.err; cat $O/bench_mt.json
timeout 300 python bench.py --model musetalk --fp8 --steps 6 --warmup 2 > $O/bench_mt_fp8.json 2> $O/bench_mt_fp8.err; cat $O/bench_mt_fp8.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/mt_trace -o r -- python $R/bench.py --model musetalk --steps 2 --warmup 1 > $O/mt_trace.log 2>&1
BCMD="python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o r -- $BCMD > $O/trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o r -- $BCMD > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o r -- $BCMD > $O/pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/pmc_sq -o r -- $BCMD > $O/pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum -d $O/pmc_l2 -o r -- $BCMD > $O/pmc_l2.log 2>&1
ls $O
