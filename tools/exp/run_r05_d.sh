#!/bin/bash
# round 5, fourth box: frame-chunked CSM on two streams (A/B), staged FIR stores (A/B), the whole GPU suite
set -u
export TMPDIR=/tmp
for rep in 1 2; do
  for k in 1 2 4 8; do
    echo "##### csm chunks $k rep $rep"; DSPTOOLBOX_AMD_CSM_CHUNKS=$k timeout -k 10 200 python3 bench.py --workload csm --steps 400 --warmup 20 --no-cpu-baseline --steady-steps 0 || exit 1
  done
  echo "##### fir bank, direct stores rep $rep"; timeout -k 10 200 python3 bench.py --workload fir_bank --steps 100 --warmup 10 --no-cpu-baseline --steady-steps 0 || exit 1
  echo "##### fir bank, staged stores rep $rep"; DSPTOOLBOX_AMD_FIR_STAGE=1 timeout -k 10 200 python3 bench.py --workload fir_bank --steps 100 --warmup 10 --no-cpu-baseline --steady-steps 0 || exit 1
done
echo "##### csm parity (chunks 4)"; timeout -k 10 300 python3 bench.py --workload csm --steps 50 --warmup 5 --steady-steps 0 | cut -c1-1500
echo "##### fir parity staged"; DSPTOOLBOX_AMD_FIR_STAGE=1 timeout -k 10 300 python3 -m pytest tests -m gpu -x -q -k "fir_bank_4097 or fir_golden or fir_one_and_two" 2>&1 | tail -3
echo "##### all gpu tests"; timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > gpurun_out/r05_d_tests.log 2>&1; tail -8 gpurun_out/r05_d_tests.log
echo done
