#!/bin/bash
# round 5: the 130-channel matrix case of fuzz_csm (seed 12) replayed, on the group kernels and on the generic kernel; the cap test
set -u
export TMPDIR=/tmp
timeout -k 10 500 python3 tools/csm_groups_case.py 2>&1 | tail -30
echo "##### generic kernel"
DSPTOOLBOX_AMD_CSM_GENERIC=1 timeout -k 10 500 python3 tools/csm_groups_case.py 2>&1 | tail -30
echo "##### cap test, fused uploads"
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sweep_shapes_over or fused_float64_upload" 2>&1 | tail -30
echo done
