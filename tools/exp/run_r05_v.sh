#!/bin/bash
# round 5: k_fir3 as the default two-partition FIR kernel: the whole GPU suite, then the FIR bank and the default line
# profiled again (r05c: the traced default line carries the FIR entry)
set -u
export TMPDIR=/tmp
echo "##### all gpu tests"
timeout -k 10 900 python3 -m pytest tests -q -m gpu -x 2>&1 | tail -6 || exit 1
echo "##### profiles (r05c: fir_bank, welch_h1)"
bash tools/prof_all.sh r05c "fir_bank welch_h1" > gpurun_out/r05_v_prof.log 2>&1
tail -3 gpurun_out/r05_v_prof.log
for W in welch_h1 fir_bank; do cp profiles/r05c_${W}_rocprofv3_summary.txt gpurun_out/r05c_${W}_summary.txt; done
echo "##### profile tests on this box"
timeout -k 10 300 python3 -m pytest tests/test_profiles.py -q 2>&1 | tail -5
echo "##### fuzz_fir 150 seed 203"; timeout -k 10 200 python3 tests/sweeps/fuzz_fir.py 150 203 2>&1 | tail -4
echo done
