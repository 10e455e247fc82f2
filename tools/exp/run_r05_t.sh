#!/bin/bash
# round 5: a second pass of the randomized sweeps with new seeds on the final library
set -u
export TMPDIR=/tmp
t0=$(date +%s)
echo "##### fuzz_parity 300 seed 101"; timeout -k 10 420 python3 tests/sweeps/fuzz_parity.py 300 101 2>&1 | tail -12; echo "[t $(( $(date +%s) - t0 )) s]"
echo "##### fuzz_csm 100 seed 102"; timeout -k 10 200 python3 tests/sweeps/fuzz_csm.py 100 102 2>&1 | tail -5; echo "[t $(( $(date +%s) - t0 )) s]"
echo "##### fuzz_fir 120 seed 103"; timeout -k 10 150 python3 tests/sweeps/fuzz_fir.py 120 103 2>&1 | tail -4; echo "[t $(( $(date +%s) - t0 )) s]"
echo "##### fuzz_misc 150 seed 104"; timeout -k 10 150 python3 tests/sweeps/fuzz_misc.py 150 104 2>&1 | tail -5; echo "[t $(( $(date +%s) - t0 )) s]"
echo "##### fuzz_api2 100 seed 105"; timeout -k 10 150 python3 tests/sweeps/fuzz_api2.py 100 105 2>&1 | tail -4; echo "[t $(( $(date +%s) - t0 )) s]"
echo done
