#!/bin/bash
# round 5, third box: device-resident API tests and timing
set -u
export TMPDIR=/tmp
echo "##### tests"; timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "device_resident or transfer_function_golden or stft_golden or istft or fir_golden or deconvolve_golden" > gpurun_out/r05_c_tests.log 2>&1; tail -40 gpurun_out/r05_c_tests.log
echo "##### device-resident API"; timeout -k 10 400 python3 tools/time_api_resident.py 2>&1 | tail -30
echo done
