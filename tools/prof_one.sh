#!/bin/bash
# rocprofv3 kernel trace + PMC passes for ONE bench workload (run on the GPU box).
# usage: tools/prof_one.sh <workload> <tag>
set -u
W=${1:-welch_h1}
TAG=${2:-dev}
export TMPDIR=/tmp
OUT=gpurun_out/prof_${TAG}_$W
mkdir -p $OUT
CMD="python3 bench.py --workload $W --steps 10 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA --output-format csv -d $OUT/pmc1 -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc2 -- $CMD > $OUT/pmc2.log 2>&1
python3 tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
