#!/bin/bash
# round 5: the randomized parity sweeps over the kernels that changed (2048-sample Welch windows, deconvolution, CSM finish)
set -u
export TMPDIR=/tmp
echo "##### fuzz_parity 400 cases seed 55"; timeout -k 10 500 python3 tests/sweeps/fuzz_parity.py 400 55 2>&1 | tail -12
echo "##### fuzz_misc"; timeout -k 10 300 python3 tests/sweeps/fuzz_misc.py 150 56 2>&1 | tail -8
echo "##### fuzz_api3"; timeout -k 10 300 python3 tests/sweeps/fuzz_api3.py 120 57 2>&1 | tail -8
echo done
