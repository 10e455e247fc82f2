#!/bin/bash
# round 5: cost of a larger byte cap of the float64 route (tools/x64_cap_time.py)
set -u
export TMPDIR=/tmp
timeout -k 10 500 python3 tools/x64_cap_time.py 2>&1 | tail -45
echo done
