#!/bin/bash
# One entry point for the jobs sent to the GPU box (`gpurun -- 'bash scripts/gpu_job.sh <mode> ...'`); everything lands under
# gpurun_out/.  Modes:
#   tests [pytest args]            the -m gpu suite (default: all of tests/), compact log
#   bench [bench.py args]          one bench line -> gpurun_out/bench_<TAG>.json
#   profile <tag>                  default bench + rocprofv3 kernel trace + the PMC passes (separate runs) of the Wav2Lip 16-frame,
#                                  256-frame and MuseTalk (fp16 / fp8) workloads -> gpurun_out/<tag>/ ; summarise with make_profile_summary.py
#   layers <knob-settings...> -- <frames...>   scripts/layer_times.py: per-layer times under knob settings, interleaved rounds
#   ab-lib <tag> <a.so> <b.so> ... in-job A/B of several builds of libltk_hip.so (LTK_LIB), two interleaved rounds (MT=0 skips MuseTalk)
#   ablate [small|big]             conv3 ablation masks per layer (needs ab_libs/libltk_hip_ablate.so: scripts/build_variant.sh ablate -DLTK_ABLATE_BUILD=1)
#   pmc-layer                      SQ counters of single-layer launches (conv_ablate.py under rocprofv3 --pmc)
#   clock [layers...]              scripts/conv_clock.py: the real conv3 kernel on random / constant / zero operands, un-profiled (HIP events) and under
#                                  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_* (one run per fill) -> gpurun_out/clock_report.txt
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out; mkdir -p $O; cd $R
MODE=$1; shift
case $MODE in
tests)
  ARGS=${*:-tests}
  timeout 1500 python -m pytest $ARGS -m gpu ${MAXFAIL:--x} -q -s 2>&1 | grep -E "^\.*\[|passed|failed|Error|error|FAIL|assert" | grep -v "^\[layer\]\|^\[mt\] [a-z_]*\.[a-z_0-9.]* " > $O/pytest_${TAG:-gpu}.log
  tail -40 $O/pytest_${TAG:-gpu}.log ;;
bench)
  timeout 600 python bench.py "$@" > $O/bench_${TAG:-default}.json 2> $O/bench_${TAG:-default}.err; tail -c 1500 $O/bench_${TAG:-default}.json ;;
profile)
  TAG=${1:-r03}; P=$O/$TAG; mkdir -p $P
  # ONLY="w2l": re-profile just the named workloads (no default bench run), e.g. after a change that touches one of them
  ONLY=${ONLY:-}
  want() { [ -z "$ONLY" ] || [[ " $ONLY " == *" $1 "* ]]; }
  if [ -z "$ONLY" ]; then timeout 900 python bench.py --steps 50 --warmup 5 > $P/bench_default.json 2> $P/bench_default.err; head -c 400 $P/bench_default.json; echo; fi
  cd /tmp && export TMPDIR=/tmp
  prof() {   # prof <name> <bench args...>: kernel trace + FETCH / WRITE / SQ / L2 counter passes, each its own run
    local n=$1; shift; want $n || return 0; local CMD="python $R/bench.py $* --no-cpu-baseline --no-also --no-traffic"
    timeout 400 rocprofv3 --kernel-trace --stats -d $P/${n}_trace -o r -- $CMD > $P/${n}_trace.log 2>&1
    # counter passes: the same launches issued one by one (LTK_GRAPH=0), so every dispatch is a plain kernel packet for the profiler
    export LTK_GRAPH=0
    timeout 400 rocprofv3 --pmc FETCH_SIZE -d $P/${n}_pmc_fetch -o r -- $CMD > $P/${n}_pmc_fetch.log 2>&1
    timeout 400 rocprofv3 --pmc WRITE_SIZE -d $P/${n}_pmc_write -o r -- $CMD > $P/${n}_pmc_write.log 2>&1
    timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $P/${n}_pmc_sq -o r -- $CMD > $P/${n}_pmc_sq.log 2>&1
    timeout 400 rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum -d $P/${n}_pmc_l2 -o r -- $CMD > $P/${n}_pmc_l2.log 2>&1
    unset LTK_GRAPH
  }
  # MuseTalk twice: the weight upload of the model load shows up as ~1100 __amd_rocclr_copyBuffer / ~470 fillBufferAligned launches
  # whatever the number of passes (mt: 3 passes, mt12: 12 passes) - they are not part of a pass
  prof w2l --steps 6 --warmup 2 --no-whole-pass --sustain 0
  prof w2l256 --sessions 16 --steps 3 --warmup 1 --sustain 0
  prof mt --model musetalk --steps 2 --warmup 1
  prof mtfp8 --model musetalk --fp8 --sessions 4 --steps 2 --warmup 1
  want mt12 && timeout 400 rocprofv3 --kernel-trace --stats -d $P/mt12_trace -o r -- python $R/bench.py --model musetalk --steps 11 --warmup 1 --no-cpu-baseline > $P/mt12_trace.log 2>&1
  cd $R; S=$O/${TAG}_summary; mkdir -p $S
  want w2l && python scripts/make_profile_summary.py $P $S/$TAG --name w2l --frames 16 --cmd "bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-also --no-traffic  (wav2lip256, 1 session, 16-frame batch)" > /dev/null
  want w2l256 && python scripts/make_profile_summary.py $P $S/$TAG --name w2l256 --frames 256 --cmd "bench.py --sessions 16 --steps 3 --warmup 1 --no-cpu-baseline --no-also --no-traffic  (16 sessions, 256-frame passes)" > /dev/null
  want mt && python scripts/make_profile_summary.py $P $S/$TAG --name mt --all-kernels --frames 16 --cmd "bench.py --model musetalk --steps 2 --warmup 1 --no-cpu-baseline" > /dev/null
  want mtfp8 && python scripts/make_profile_summary.py $P $S/$TAG --name mtfp8 --all-kernels --frames 64 --cmd "bench.py --model musetalk --fp8 --sessions 4 --steps 2 --warmup 1 --no-cpu-baseline" > /dev/null
  want mt12 && python scripts/prof_report.py $P/mt12_trace 2>/dev/null | grep -E "^kernel|amd_rocclr|gn_|conv3_kernel<1, 2, 4" | head -12 > $S/${TAG}_mt12_blits.txt
  [ -z "$ONLY" ] && cp $P/bench_default.json $S/${TAG}_bench_default.json
  rm -rf $P
  ls $P ;;
layers)
  ROUNDS=${ROUNDS:-3} timeout 900 python scripts/layer_times.py "$@" 2>&1 | tee $O/layers_${TAG:-run}.txt | tail -70 ;;
ab-lib)
  TAG=$1; shift; LOG=$O/ab_$TAG.log; : > $LOG
  for rnd in 1 2; do for lib in "$@"; do
    echo "######## round $rnd lib $lib" >> $LOG
    LTK_LIB=$R/$lib ROUNDS=3 timeout 300 python scripts/layer_times.py "TILE_RULE=1" -- ${FRAMES:-16 256} 2>&1 | grep -E "${ROWS:-^(====|sum|conv stack|face_decoder_blocks.[3-7]|face_encoder_blocks.[2-6].[01]|output)}" >> $LOG
    if [ "${MT:-1}" != "0" ]; then
      LTK_LIB=$R/$lib timeout 300 python scripts/mt_op_times.py 16 2>&1 | grep -E "pass|conv/linear|GroupNorm|resnets.1.conv2|resnets.1.conv1" | head -8 >> $LOG
    fi
  done; done
  cat $LOG ;;
ablate)
  LTK_LIB=$R/ab_libs/libltk_hip_ablate.so ABLATE_SET=$1 SWEEP_FRAMES=${FRAMES:-16} ABLATE_MASKS=${MASKS:-0,1,2,3,4,7,16,64,68} timeout 900 python scripts/conv_ablate.py 2>&1 | tee $O/ablate_${TAG:-run}.txt ;;
pmc-layer)
  P=$O/pmc_layer; mkdir -p $P; cd /tmp; export TMPDIR=/tmp
  export LTK_LIB=$R/ab_libs/libltk_hip_ablate.so ABLATE_MASKS=${MASKS:-0} ABLATE_SET=${SET:-big}
  CMD="python $R/scripts/conv_ablate.py"
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $P/a -o r -- $CMD > $P/a.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_MISC -d $P/b -o r -- $CMD > $P/b.log 2>&1
  rocprofv3 --kernel-trace -d $P/t -o r -- $CMD > $P/t.log 2>&1
  tail -3 $P/b.log ;;
clock)
  P=$O/clock; rm -rf $P; mkdir -p $P; LAYERS=${*:-c256@64 c64@256}
  for L in $LAYERS; do for F in random constant zeros random; do timeout 120 python scripts/conv_clock.py $F $L 200 2>&1 | grep "^\[clock\]" >> $P/unprofiled.log; done; done
  cd /tmp; export TMPDIR=/tmp
  for L in $LAYERS; do for F in random constant zeros; do
    timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 \
      -d $P/${L}_$F -o r -- python $R/scripts/conv_clock.py $F $L 200 > $P/${L}_$F.log 2>&1
  done; done
  cd $R; python scripts/clock_report.py $P > $O/clock_report.txt 2>&1; cat $P/unprofiled.log; cat $O/clock_report.txt
  find $P -name "*.db" -size +20M -delete ;;
r4a)  # round-4 job A: full GPU suite, graph / depth-first A/B, clock experiment, short bench lines, scheduler in-flight A/B
  TAG=r4a bash $0 tests tests > /dev/null 2>&1; tail -5 $O/pytest_r4a.log
  ROUNDS=3 timeout 600 python scripts/layer_times.py "GRAPH=0" "GRAPH=1" "GRAPH=1,DF_FRAMES=4,DF_MIN=1" "GRAPH=1,DF_FRAMES=8,DF_MIN=1" "GRAPH=1,DF_FRAMES=16,DF_MIN=1" \
      "GRAPH=1,DF_FRAMES=4,DF_MIN=1,DF_BLOCK=7" "GRAPH=1,DF_FRAMES=8,DF_MIN=1,DF_BLOCK=5" -- 16 64 256 2>&1 | grep -E "^====|^sum|^conv stack" > $O/r4a_graph_df_ab.txt; cat $O/r4a_graph_df_ab.txt
  bash $0 clock c256@64 c64@256 > $O/r4a_clock.log 2>&1; cat $O/clock_report.txt
  timeout 400 python bench.py --steps 50 --warmup 5 --no-also --no-cpu-baseline > $O/r4a_bench_s1.json 2> $O/r4a_bench_s1.err; head -c 1200 $O/r4a_bench_s1.json; echo
  for IF in 1 2 1 2; do LTK_INFLIGHT=$IF timeout 300 python bench.py --sessions 16 --steps 10 --warmup 3 --no-also --no-cpu-baseline --no-traffic > $O/r4a_bench_s16_if$IF.json 2>> $O/r4a_bench_s16.err
    python -c "import json,sys; d=json.load(open('$O/r4a_bench_s16_if$IF.json')); print('inflight $IF', d['value'], d['ms_per_step'], d['roofline']['frac'], d['scheduler'])"; done ;;
r4b)  # round-4 job B: paste diagnostic, the whole GPU suite (no stop at the first failure), depth-first / stagger / rowconv A/Bs, graph A/B of the timed line
  timeout 200 python scripts/paste_diag.py 2>&1 | grep "paste-diag" > $O/r4b_paste_diag.txt; cat $O/r4b_paste_diag.txt
  TAG=r4b MAXFAIL=--maxfail=20 bash $0 tests tests > /dev/null 2>&1; grep -E "passed|failed|FAILED" $O/pytest_r4b.log | tail -12
  ROUNDS=7 timeout 500 python scripts/pass_ab.py "DF_FRAMES=0" "DF_FRAMES=16,DF_MIN=1" "DF_FRAMES=12,DF_MIN=1" "DF_FRAMES=24,DF_MIN=1" "DF_FRAMES=32,DF_MIN=1" "DF_FRAMES=16,DF_MIN=1,DF_BLOCK=5" "DF_FRAMES=16,DF_MIN=1,DF_BLOCK=7" "DF_FRAMES=16,DF_MIN=1,DF_BLOCK=4" -- 32 48 64 128 256 > $O/r4b_df_ab.txt 2>&1; cat $O/r4b_df_ab.txt
  ROUNDS=7 timeout 500 python scripts/pass_ab.py "STAGGER=0" "STAGGER=3000" "STAGGER=8000" "STAGGER=16000" "STAGGER=30000" "STAGGER=8000,STAGGER_MODE=1" "STAGGER=16000,STAGGER_MODE=1" -- 16 64 256 > $O/r4b_stagger_ab.txt 2>&1; cat $O/r4b_stagger_ab.txt
  ROUNDS=7 timeout 300 python scripts/pass_ab.py "ROWCONV=1024" "ROWCONV=2048" "ROWCONV=0" -- 16 32 > $O/r4b_rowconv_ab.txt 2>&1; cat $O/r4b_rowconv_ab.txt
  for G in 0 1 0 1; do LTK_GRAPH=$G timeout 300 python bench.py --steps 100 --warmup 5 --no-also --no-cpu-baseline --no-traffic > $O/r4b_bench_g$G.json 2>> $O/r4b_bench.err
    python -c "import json; d=json.load(open('$O/r4b_bench_g$G.json')); print('graph $G', d['value'], d['ms_per_step'], d['roofline']['conv_stack_ms'], d['roofline']['hipgraph'], d['pcie_inclusive']['value'])"; done ;;
r4c)  # round-4 job C: per-phase rowconv for the small-map transposed convs
  TAG=r4c MAXFAIL=--maxfail=20 bash $0 tests tests/test_wav2lip_gpu.py tests/test_mel_paste_gpu.py tests/test_plugin_gpu.py > /dev/null 2>&1; grep -E "passed|failed|FAILED|rowconv" $O/pytest_r4c.log | tail -16
  ROUNDS=9 timeout 300 python scripts/pass_ab.py "ROWCONVT=0" "ROWCONVT=1" -- 16 8 4 > $O/r4c_rowconvT_ab.txt 2>&1; cat $O/r4c_rowconvT_ab.txt
  ROUNDS=3 timeout 300 python scripts/layer_times.py "ROWCONVT=0" "ROWCONVT=1" -- 16 2>&1 | grep -E "^====|face_decoder_blocks.[123]|^sum|^conv stack" > $O/r4c_rowconvT_layers.txt; cat $O/r4c_rowconvT_layers.txt ;;
r4d)  # round-4 job D: conv3 stride-2 with the conflict-free LDS image (default routing and LTK_CONV_V3_S2=2 = every stride-2 3x3 layer on conv3); scheduler in-flight A/B
  TAG=r4d MAXFAIL=--maxfail=20 bash $0 tests tests/test_conv_gpu.py tests/test_wav2lip_gpu.py tests/test_musetalk_gpu.py > /dev/null 2>&1; grep -E "passed|failed|FAILED" $O/pytest_r4d.log | tail -8
  LTK_CONV_V3_S2=2 TAG=r4d_s2all MAXFAIL=--maxfail=20 bash $0 tests tests/test_conv_gpu.py tests/test_wav2lip_gpu.py > /dev/null 2>&1; grep -E "passed|failed|FAILED" $O/pytest_r4d_s2all.log | tail -8
  for rnd in 1 2; do for V in 1 2; do
    echo "######## round $rnd LTK_CONV_V3_S2=$V" >> $O/r4d_s2_layers.txt
    LTK_CONV_V3_S2=$V ROUNDS=3 timeout 300 python scripts/layer_times.py "GRAPH=1" -- 16 256 2>&1 | grep -E "^====|face_encoder_blocks.[1-6].0|^sum|^conv stack" >> $O/r4d_s2_layers.txt
  done; done; cat $O/r4d_s2_layers.txt
  for IF in 1 2 1 2 1 2; do LTK_INFLIGHT=$IF timeout 300 python bench.py --sessions 16 --steps 12 --warmup 3 --no-also --no-cpu-baseline --no-traffic > $O/r4d_bench_s16_if$IF.json 2>> $O/r4d_bench_s16.err
    python -c "import json,sys; d=json.load(open('$O/r4d_bench_s16_if$IF.json')); print('inflight $IF', d['value'], d['ms_per_step'], d['roofline']['frac'], d['scheduler'])" | tee -a $O/r4d_inflight.txt; done ;;
r4e)  # round-4 job E: concurrency test, delivered capacity with / without the graph path, the MuseTalk sub-bench with its measured delivered run
  TAG=r4e MAXFAIL=--maxfail=20 bash $0 tests tests/test_plugin_gpu.py tests/test_mel_paste_gpu.py > /dev/null 2>&1; grep -E "passed|failed|FAILED|in flight" $O/pytest_r4e.log | tail -8
  for G in 0 1 0 1; do LTK_GRAPH=$G timeout 400 python bench.py --sub delivered-capacity --batch 16 --delivered-sessions 384,448,512 > $O/r4e_delivered_g$G.json 2>> $O/r4e_delivered.err
    python -c "
import json; d=json.load(open('$O/r4e_delivered_g$G.json'))
for f in ('bgr24','i420'): print('graph $G', f, [(t['sessions'], t.get('latency_ms_max'), t.get('sustained')) for t in d[f]['tested']])" | tee -a $O/r4e_delivered.txt; done
  timeout 900 python bench.py --sub musetalk-both --batch 16 > $O/r4e_musetalk_both.json 2> $O/r4e_musetalk_both.err; python -c "
import json; d=json.load(open('$O/r4e_musetalk_both.json'))
for o in d: print(o.get('value'), o.get('roofline',{}).get('frac'), o.get('sessions_25fps'), json.dumps(o.get('delivered'))[:900])" ;;
r4g)  # round-4 job G: MuseTalk small-map linear layers on rowconv
  TAG=r4g MAXFAIL=--maxfail=20 bash $0 tests tests/test_musetalk_gpu.py tests/test_musetalk_plugin_gpu.py tests/test_fp8_gpu.py tests/test_wav2lip_gpu.py > /dev/null 2>&1; grep -E "passed|failed|FAILED" $O/pytest_r4g.log | tail -8
  timeout 500 python scripts/mt_op_times.py 16 MT_ROWCONV=0,1024 2>&1 | grep -E "^====|conv/linear|GroupNorm|by block|->" > $O/r4g_mt_rowconv_ab.txt; head -60 $O/r4g_mt_rowconv_ab.txt
  timeout 400 python scripts/mt_op_times.py 64 MT_ROWCONV=0,1024 2>&1 | grep -E "^====|conv/linear|->" > $O/r4g_mt_rowconv_ab64.txt; head -30 $O/r4g_mt_rowconv_ab64.txt ;;
r4h)  # round-4 job H: MuseTalk per-level tile width of the U-Net's 3x3 convs
  TAG=r4h MAXFAIL=--maxfail=20 bash $0 tests tests/test_musetalk_gpu.py tests/test_musetalk_plugin_gpu.py > /dev/null 2>&1; grep -E "passed|failed|FAILED" $O/pytest_r4h.log | tail -8
  timeout 500 python scripts/mt_op_times.py 16 MT_TILE_TABLE=0,1 2>&1 | grep -E "^====|conv/linear|by block|->" > $O/r4h_mt_tile_ab.txt; grep -E "^====|->" $O/r4h_mt_tile_ab.txt | head -70 ;;
r4i)  # round-4 job I: 128-cout blocks + forced split-K for the small-map linear layers of MuseTalk (conv3 1x1), per-op response
  ALL_OPS=attentions,conv_shortcut timeout 500 python scripts/mt_op_times.py 16 MT_ROWCONV=1024,0,0+CONV3_NBT=0,4,4+KSPLIT=0,8,4 2>&1 | grep -E "^====| us  conv/linear" | grep -E "^====|down_blocks.2|mid_block|up_blocks.[01]" > $O/r4i_mt_nbt4_ops.txt; head -150 $O/r4i_mt_nbt4_ops.txt ;;
r4m)  # round-4 job M: fused small-map GroupNorm (MuseTalk)
  TAG=r4m MAXFAIL=--maxfail=20 bash $0 tests tests/test_musetalk_gpu.py tests/test_musetalk_plugin_gpu.py tests/test_fp8_gpu.py > /dev/null 2>&1; grep -E "passed|failed|FAILED" $O/pytest_r4m.log | tail -8
  timeout 500 python scripts/mt_op_times.py 16 MT_GN_FUSED=0,1 2>&1 | grep -E "^====|GroupNorm|->" > $O/r4m_mt_gn_fused_ab.txt; grep -E "^====|GroupNorm " $O/r4m_mt_gn_fused_ab.txt | head; grep -E "\->" $O/r4m_mt_gn_fused_ab.txt | head -70 ;;
r4vs3)  # the round-3 tree (build/r03tree: git archive 367ff43 + make) against this tree in ONE job, processes interleaved
  L=$O/r4vs3.txt; : > $L
  for rnd in 1 2 3; do for T in r03 r04; do
    D=$R; [ $T = r03 ] && D=$R/build/r03tree
    echo "######## round $rnd tree $T: conv stack / pass (scripts/layer_times.py, 3 rounds; us)" >> $L
    (cd $D && ROUNDS=3 timeout 300 python scripts/layer_times.py "TILE_RULE=1" -- 16 256 2>&1 | grep -E "^====|^sum|^conv stack") >> $L
    echo "######## round $rnd tree $T: timed line (bench.py --steps 100, 1 session x 16 frames)" >> $L
    (cd $D && timeout 300 python bench.py --steps 100 --warmup 5 --no-also --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['conv_stack_ms'], d['roofline']['frac'])") >> $L
  done; done
  for rnd in 1 2; do for T in r03 r04; do
    D=$R; [ $T = r03 ] && D=$R/build/r03tree
    echo "######## round $rnd tree $T: MuseTalk 16-frame pass (scripts/mt_op_times.py 16)" >> $L
    (cd $D && timeout 300 python scripts/mt_op_times.py 16 2>&1 | grep -E "^==== |conv/linear|GroupNorm ") >> $L
  done; done
  cat $L ;;
r4n)  # round-4 job N: the final default (GRAPH = 1: replay from 48 frames on) - tests that touch the graph path, timed line auto vs always, the default bench line
  TAG=r4n MAXFAIL=--maxfail=20 bash $0 tests tests/test_wav2lip_gpu.py tests/test_plugin_gpu.py > /dev/null 2>&1; grep -E "passed|failed|FAILED" $O/pytest_r4n.log | tail -6
  for G in 1 2 1 2 1 2; do LTK_GRAPH=$G timeout 300 python bench.py --steps 100 --warmup 5 --no-also --no-cpu-baseline --no-traffic > $O/r4n_bench_g$G.json 2>> $O/r4n_bench.err
    python -c "import json; d=json.load(open('$O/r4n_bench_g$G.json')); print('GRAPH=$G', d['value'], d['ms_per_step'], d['roofline']['conv_stack_ms'], d['roofline']['frac'], d['roofline']['hipgraph'])" | tee -a $O/r4n_graph_auto_ab.txt; done
  timeout 600 python bench.py > $O/r4n_bench_default.json 2> $O/r4n_bench_default.err; head -c 300 $O/r4n_bench_default.json ;;
r4p)  # round-4 job P: row-parity LDS key on tiles narrower than 32 pixels (conv3 stride 1): in-job A/B first, then the whole -m gpu suite, smoke, a bench line
  L=$O/r4p_lds_swz_ab.txt; : > $L
  ROUNDS=5 timeout 300 python scripts/pass_ab.py "LDS_SWZ=0" "LDS_SWZ=1" -- 16 64 256 2>&1 | grep -E "^settings|frames" >> $L
  timeout 400 python scripts/mt_op_times.py 16 LDS_SWZ=0,1 2>&1 | grep -E "^====|conv/linear|->" > $O/r4p_mt_lds_swz_ab.txt; grep -E "^====|conv/linear " $O/r4p_mt_lds_swz_ab.txt | head -12 >> $L; cat $L
  TAG=r4p MAXFAIL=--maxfail=20 bash $0 tests tests > /dev/null 2>&1; grep -E "passed|failed|FAILED" $O/pytest_r4p.log | tail -8
  timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -2
  timeout 400 python bench.py --no-cpu-baseline > $O/r4p_bench.json 2> $O/r4p_bench.err; head -c 600 $O/r4p_bench.json ;;
r4q)  # round-4 job Q: SQ_LDS_BANK_CONFLICT of the MuseTalk pass (then the 256-frame Wav2Lip pass) under the column key and the row key (separate --pmc runs, eager launches)
  cd /tmp && export TMPDIR=/tmp; export LTK_GRAPH=0
  for W in mt w2l256; do
    [ $W = mt ] && A="--model musetalk --steps 2 --warmup 1" || A="--sessions 16 --steps 2 --warmup 1"
    for K in 0 1; do LTK_LDS_SWZ=$K timeout 130 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/r4q/${W}_swz$K -o r -- python $R/bench.py $A --no-cpu-baseline --no-also --no-traffic > $O/r4q_${W}_swz$K.log 2>&1; done
    python $R/scripts/lds_conflict_report.py column-key=$O/r4q/${W}_swz0 row-key=$O/r4q/${W}_swz1 > $O/r4q_lds_conflicts_$W.txt 2>&1; head -30 $O/r4q_lds_conflicts_$W.txt
    rm -rf $O/r4q/${W}_swz0 $O/r4q/${W}_swz1
  done ;;
r4r)  # round-4 job R: host-path trims (unbind, lazy bank indices, scheduler solo path): the plugin-level GPU tests, then the timed line (gap = ms_per_step - device pass)
  TAG=r4r MAXFAIL=--maxfail=20 timeout 120 bash $0 tests tests/test_plugin_gpu.py tests/test_egress_gpu.py tests/test_ref_loop.py > /dev/null 2>&1; grep -E "passed|failed|FAILED|rror" $O/pytest_r4r.log | tail -8
  for i in 1 2; do timeout 40 python bench.py --steps 200 --warmup 10 --no-also --no-cpu-baseline --no-traffic 2>> $O/r4r_bench.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('timed line', d['value'], d['ms_per_step'], d['roofline']['conv_stack_ms'], round(1e3*(d['ms_per_step']-d['roofline']['conv_stack_ms']),1), 'us host gap')" | tee -a $O/r4r_host_gap.txt; done
  TAG=r4r_mt MAXFAIL=--maxfail=20 timeout 100 bash $0 tests tests/test_musetalk_plugin_gpu.py > /dev/null 2>&1; grep -E "passed|failed|FAILED|rror" $O/pytest_r4r_mt.log | tail -4 ;;
r4s)  # round-4 job S: host-path trims, in-job A/B of the Python trees (build/oldhost = git archive d39d0bc + the same libltk_hip.so), 16 sessions and 1 session
  L=$O/r4s_host_ab.txt; : > $L
  one() { (cd $1 && timeout 40 python bench.py --sessions $2 --steps $3 --warmup 5 --no-also --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$4 sessions=$2', d['value'], d['ms_per_step'])") | tee -a $L; }
  for rnd in 1 2; do one $R/build/oldhost 16 40 old; one $R 16 40 new; done
  one $R/build/oldhost 1 200 old; one $R 1 200 new ;;
r4t)  # round-4 job T: the 1x1 chunk ring (LTK_RING1; needs the tree of commit cb1ad49): bitwise parity + per-shape timing, then the MuseTalk pass per op
  timeout 35 python scripts/ring1_ab.py quick > $O/r4t_ring1_shapes.txt 2>&1; tail -12 $O/r4t_ring1_shapes.txt
  timeout 45 python scripts/mt_op_times.py 16 RING1=0,4 2>&1 | grep -E "^====|conv/linear  |->" > $O/r4t_mt_ring1.txt; grep -E "^====" $O/r4t_mt_ring1.txt; grep -E "\->" $O/r4t_mt_ring1.txt | head -30 ;;
*) echo "unknown mode $MODE"; exit 2 ;;
esac
