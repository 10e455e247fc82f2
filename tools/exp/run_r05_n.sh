#!/bin/bash
# round 5: the coherence of a bin the output barely excites (tools/nyquist_null.py), register kernels and generic kernels
set -u
export TMPDIR=/tmp
echo "##### 4096, register machine"; timeout -k 10 300 python3 tools/nyquist_null.py 4096 2>&1 | tail -20
echo "##### 4096, generic LDS kernels"; DSPTOOLBOX_AMD_NO_WELCH4096=1 DSPTOOLBOX_AMD_WELCH_GENERIC=1 timeout -k 10 300 python3 tools/nyquist_null.py 4096 2>&1 | tail -20
echo "##### 1024, wave kernels"; timeout -k 10 300 python3 tools/nyquist_null.py 1024 2>&1 | tail -20
echo done
