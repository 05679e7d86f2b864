#!/bin/bash
# PMC on single-layer launches (conv_ablate.py with given masks)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_layer; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
export ABLATE_MASKS=${ABLATE_MASKS:-0,27}
CMD="python $R/scripts/conv_ablate.py"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/a -o r -- $CMD > $O/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_MISC -d $O/b -o r -- $CMD > $O/b.log 2>&1
rocprofv3 --kernel-trace -d $O/t -o r -- $CMD > $O/t.log 2>&1
tail -3 $O/b.log
