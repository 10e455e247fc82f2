#!/bin/bash
# round 5, final check on the final library: smoke, the whole GPU suite, the two-rank rehearsal of the default workload
set -u
export TMPDIR=/tmp
echo "##### smoke"; timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "##### all gpu tests"; timeout -k 10 900 python3 -m pytest tests -m gpu -q > gpurun_out/r05_x_tests.log 2>&1; tail -4 gpurun_out/r05_x_tests.log
echo "##### two ranks on this one GPU (host broadcast), weak"
BENCH_BCAST=host timeout -k 10 400 python3 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05_x_2rank.json 2> gpurun_out/r05_x_2rank.err; echo "rc $?"; cut -c1-400 gpurun_out/r05_x_2rank.json
ls -la gpurun_out/.graft_exec_refused 2>/dev/null
echo done
