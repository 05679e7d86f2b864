R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; P=$O/r05p; mkdir -p $P; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $P/w2l_trace -o r -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-also --no-traffic --sustain 0 --no-whole-pass > $P/w2l_trace.log 2>&1
tail -2 $P/w2l_trace.log | cut -c1-600
cd $R; mkdir -p $O/r05p_summary
python scripts/make_profile_summary.py $P $O/r05p_summary/r05p --name w2l --frames 16 --cmd "bench.py --steps 6 --warmup 3 (prefetch on)" > /dev/null 2>$O/r05p_summary/err.txt; tail -3 $O/r05p_summary/err.txt
rm -rf $P; ls $O/r05p_summary
