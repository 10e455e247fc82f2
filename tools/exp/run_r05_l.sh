#!/bin/bash
# round 5: the one paired-input case of run K above 1e-6, with its bin, channel and coherence
set -u
export TMPDIR=/tmp
echo "##### fuzz_api3 120 57"; timeout -k 10 400 python3 tests/sweeps/fuzz_api3.py 120 57 2>&1 | tail -8
echo "##### fuzz_api3 200 58"; timeout -k 10 500 python3 tests/sweeps/fuzz_api3.py 200 58 2>&1 | tail -8
echo done
