#!/bin/bash
# usage (inside gpurun): bash tools/ct_trace_run.sh r03     -> gpurun_out/r03_ct_instruction_counts.json
ROUND=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/ct_trace
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE \
  -d $OUT -o pmc -- python3 $REPO/tools/ct_trace_check.py run > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
python3 $REPO/tools/ct_trace_check.py summarize $OUT > $REPO/gpurun_out/${ROUND}_ct_instruction_counts.json 2> $REPO/gpurun_out/${ROUND}_ct_instruction_counts.txt
cat $REPO/gpurun_out/${ROUND}_ct_instruction_counts.txt
rm -rf $OUT
