#!/bin/bash
# round 5: 2048-sample windows on the 4096-point machine (tests, A/B timing), the whole GPU suite
set -u
export TMPDIR=/tmp
echo "##### welch 2048 tests"; timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "welch_2048 or device_resident or switch" > gpurun_out/r05_f_w2048_tests.log 2>&1; tail -25 gpurun_out/r05_f_w2048_tests.log
for rep in 1 2; do
  echo "##### welch sizes (default routes) rep $rep"; timeout -k 10 300 python3 tools/time_welch_sizes.py 256 1024 2048 4096 8192 16384 2>&1 | grep nfft
  echo "##### welch 2048 on the wave kernels rep $rep"; DSPTOOLBOX_AMD_W2048_WAVE=1 timeout -k 10 300 python3 tools/time_welch_sizes.py 2048 2>&1 | grep nfft
done
echo "##### all gpu tests"; timeout -k 10 1000 python3 -m pytest tests -m gpu -q > gpurun_out/r05_f_tests.log 2>&1; tail -12 gpurun_out/r05_f_tests.log
echo done
