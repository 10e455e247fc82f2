#!/bin/bash
# round 5: the whole GPU suite on the final library, then the sweeps that run K did not cover
set -u
export TMPDIR=/tmp
echo "##### all gpu tests"
timeout -k 10 900 python3 -m pytest tests -q -m gpu -x 2>&1 | tail -6 || exit 1
echo "##### fuzz_csm"; timeout -k 10 300 python3 tests/sweeps/fuzz_csm.py 60 12 2>&1 | tail -4
echo "##### fuzz_fir"; timeout -k 10 300 python3 tests/sweeps/fuzz_fir.py 80 13 2>&1 | tail -4
echo "##### fuzz_api2"; timeout -k 10 300 python3 tests/sweeps/fuzz_api2.py 80 14 2>&1 | tail -4
echo "##### fuzz_long_windows"; timeout -k 10 400 python3 tests/sweeps/fuzz_long_windows.py 60 15 2>&1 | tail -4
echo done
