#!/bin/bash
set -u
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "csm or das or beamform" 2>&1 | tail -3
echo done
