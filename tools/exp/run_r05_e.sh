#!/bin/bash
# round 5: device-resident API (tests + timing), frame-chunked CSM (A/B), staged FIR stores (A/B), the whole GPU suite
set -u
export TMPDIR=/tmp
echo "##### device-resident tests"; timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "device_resident" > gpurun_out/r05_e_resident_tests.log 2>&1; tail -30 gpurun_out/r05_e_resident_tests.log
echo "##### device-resident API"; timeout -k 10 400 python3 tools/time_api_resident.py > gpurun_out/r05_api_resident.txt 2>&1; tail -30 gpurun_out/r05_api_resident.txt
for rep in 1 2; do
  for k in 1 2 4 8; do
    echo "##### csm chunks $k rep $rep"; DSPTOOLBOX_AMD_CSM_CHUNKS=$k timeout -k 10 200 python3 bench.py --workload csm --steps 400 --warmup 20 --no-cpu-baseline --steady-steps 0 || exit 1
  done
  echo "##### fir bank, direct stores rep $rep"; timeout -k 10 200 python3 bench.py --workload fir_bank --steps 100 --warmup 10 --no-cpu-baseline --steady-steps 0 || exit 1
  echo "##### fir bank, staged stores rep $rep"; DSPTOOLBOX_AMD_FIR_STAGE=1 timeout -k 10 200 python3 bench.py --workload fir_bank --steps 100 --warmup 10 --no-cpu-baseline --steady-steps 0 || exit 1
done
echo "##### csm parity (chunks 4)"; timeout -k 10 300 python3 bench.py --workload csm --steps 50 --warmup 5 --steady-steps 0 | cut -c1-1500
echo "##### fir parity staged"; DSPTOOLBOX_AMD_FIR_STAGE=1 timeout -k 10 300 python3 -m pytest tests -m gpu -x -q -k "fir_bank_4097 or fir_golden or fir_one_and_two" 2>&1 | tail -3
echo "##### all gpu tests"; timeout -k 10 1000 python3 -m pytest tests -m gpu -q > gpurun_out/r05_e_tests.log 2>&1; tail -12 gpurun_out/r05_e_tests.log
echo done
