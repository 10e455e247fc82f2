#!/bin/bash
# round 5: the short-estimate sweep (1 ... 5 frames, every kind, through the API) and the long-window sweep after the byte-cap
# change and the float64 median matrix kernel
set -u
export TMPDIR=/tmp
echo "##### edge_welch"; timeout -k 10 700 python3 tests/sweeps/edge_welch.py 2>&1 | tail -12
echo "##### fuzz_long_windows 60 seed 31"; timeout -k 10 400 python3 tests/sweeps/fuzz_long_windows.py 60 31 2>&1 | tail -4
echo done
