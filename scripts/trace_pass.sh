#!/bin/bash
# Kernel trace of the timed Wav2Lip configuration only (no PMC passes): `gpurun -- bash scripts/trace_pass.sh` writes the per-kernel stats and the
# launch timeline of the last pass to gpurun_out/r05p_summary/ (scripts/make_profile_summary.py).  The quick look used while ordering the
# prefetched branch of knob PREFETCH (round 5); the full set is `scripts/gpu_job.sh profile <tag>`.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; P=$O/r05p; mkdir -p $P; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $P/w2l_trace -o r -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-also --no-traffic --sustain 0 --no-whole-pass > $P/w2l_trace.log 2>&1
tail -2 $P/w2l_trace.log | cut -c1-600
cd $R; mkdir -p $O/r05p_summary
python scripts/make_profile_summary.py $P $O/r05p_summary/r05p --name w2l --frames 16 --cmd "bench.py --steps 6 --warmup 3 (prefetch on)" > /dev/null 2>$O/r05p_summary/err.txt; tail -3 $O/r05p_summary/err.txt
rm -rf $P; ls $O/r05p_summary
