#!/bin/bash
# round 5: the whole GPU suite after the cap change / ds_welch_csd_f64 / result pool change, then the 100-frame timing rows
set -u
export TMPDIR=/tmp
echo "##### all gpu tests"
timeout -k 10 900 python3 -m pytest tests -q -m gpu -x 2>&1 | tail -15 || exit 1
echo "##### x64_cap_time"
timeout -k 10 600 python3 tools/x64_cap_time.py 2>&1 | grep -v Warn | grep -v "warn(" | tail -60
echo done
