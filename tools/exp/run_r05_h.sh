#!/bin/bash
# round 5, final check: smoke, the whole GPU suite, the driver's bench command, the spilling generic 16384-point routes timed
set -u
export TMPDIR=/tmp
echo "##### smoke"; timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "##### all gpu tests"; timeout -k 10 900 python3 -m pytest tests -m gpu -q > gpurun_out/r05_h_tests.log 2>&1; tail -6 gpurun_out/r05_h_tests.log
echo "##### driver bench"; timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_h_bench.json 2> gpurun_out/r05_h_bench.err; cut -c1-600 gpurun_out/r05_h_bench.json
echo "##### generic 16384-point routes (they spill: DESIGN section 7)"
DSPTOOLBOX_AMD_WELCH_GENERIC=1 timeout -k 10 200 python3 tools/time_welch_sizes.py 16384 2>&1 | grep nfft
timeout -k 10 200 python3 tools/time_welch_sizes.py 16384 2>&1 | grep nfft
DSPTOOLBOX_AMD_STFT_GENERIC=1 timeout -k 10 200 python3 tools/time_stft_sizes.py 16384 2>&1 | tail -2
timeout -k 10 200 python3 tools/time_stft_sizes.py 16384 2>&1 | tail -2
echo done
