#!/bin/bash
# rocprofv3 PMC + kernel-trace summaries for every bench workload (run on the GPU box).
# usage: tools/prof_all.sh <tag> ["workload ..."]        -> gpurun_out/prof_<tag>_<workload>/summary.txt
# Order: the PMC passes first, condensed into profiles/<tag>_<workload>_rocprofv3_summary.txt of the box's copy of
# the tree -- so that the traced bench run that follows reads THIS round's counters (traffic, instruction counts)
# and stamps this round's file name into its line --, then the kernel trace of the bench command itself.
set -u
TAG=${1:-r01}
export TMPDIR=/tmp
WORKLOADS=${2:-"welch_h1 welch_h1_1024 fir_bank csm deconv"}
for W in $WORKLOADS; do
  OUT=gpurun_out/prof_${TAG}_$W
  rm -rf $OUT
  mkdir -p $OUT
  # (counter passes: per-dispatch averages, no need for the clock-ramp preheat or the 2000-step steady-state region)
  CMD="python3 bench.py --workload $W --steps 10 --warmup 2 --no-cpu-baseline --preheat-ms 0 --steady-steps 0"
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA --output-format csv -d $OUT/pmc1 -- $CMD > $OUT/pmc1.log 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc2 -- $CMD > $OUT/pmc2.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -- $CMD > $OUT/pmc3.log 2>&1
  rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc4 -- $CMD > $OUT/pmc4.log 2>&1
  # issue-side evidence: cycles the vector / LDS / memory instructions hold a wave's issue, instructions in flight at
  # the LDS, scalar instruction count
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc5 -- $CMD > $OUT/pmc5.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_WAVES SQ_IFETCH SQ_INSTS_SMEM --output-format csv -d $OUT/pmc6 -- $CMD > $OUT/pmc6.log 2>&1
  python3 tools/prof_summary.py $OUT > profiles/${TAG}_${W}_rocprofv3_summary.txt 2>&1
  # kernel trace: the bench command itself -- the headline workload as the driver runs it (the default line WITH its
  # "workloads" entries), the others as their own command
  if [ "$W" = "welch_h1" ]; then TRACE="python3 bench.py"; else TRACE="python3 bench.py --workload $W"; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $TRACE > $OUT/trace.log 2>&1
  python3 tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
  grep "^{\"metric" $OUT/trace.log > $OUT/bench_line.txt
  echo "== bench line of the traced run (rocprofv3 --kernel-trace --stats -- $TRACE)" >> $OUT/summary.txt
  cat $OUT/bench_line.txt >> $OUT/summary.txt
  cp $OUT/summary.txt profiles/${TAG}_${W}_rocprofv3_summary.txt
  # gpurun merges at most 64 MiB back: the per-dispatch CSVs have been condensed into summary.txt
  rm -rf $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4 $OUT/pmc5 $OUT/pmc6 $OUT/trace
  echo "$W done"
done
echo done
