#!/bin/bash
# rocprofv3 PMC passes over the A/B harness (run on the GPU box):  tools/exp/prof_exp.sh <tag> [harness args]
set -u
TAG=${1:-dev}
shift
export TMPDIR=/tmp
OUT=gpurun_out/prof_exp_$TAG
mkdir -p $OUT
CMD="tools/exp/exp_w4 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc1 -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc2 -- $CMD > $OUT/pmc2.log 2>&1
rocprofv3 --pmc SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_IFETCH SQ_WAVES --output-format csv -d $OUT/pmc3 -- $CMD > $OUT/pmc3.log 2>&1
python3 tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
