#!/bin/bash
# round 5: "no prefetch, one more workgroup per CU": the two-partition FIR bank on k_fir3 (prefetch depth 0 / 2 / 4) against
# k_fir<2>, and the long-window Welch cross loop k_yc<false, JIT> against k_yc<false>; same box, alternating
set -u
export TMPDIR=/tmp
for rep in 1 2 3; do
  for v in 0 1 2 4; do
    echo "##### rep $rep DSPTOOLBOX_AMD_FIR_3PERCU=$v"
    DSPTOOLBOX_AMD_FIR_3PERCU=$v timeout -k 10 300 python3 bench.py --workload fir_bank --steps 40 --warmup 5 --no-cpu-baseline --steady-steps 0 > gpurun_out/r05_u_line.json 2> gpurun_out/r05_u.err || { tail -5 gpurun_out/r05_u.err; exit 1; }
    python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_u_line.json').read().strip().splitlines()[-1])
r=d['roofline']
print('ms_per_step',round(d['ms_per_step'],4),'kernel_avg_ms',round(r['kernel_avg_ms'],4),'frac',round(r['frac'],4))
PY
  done
done
for rep in 1 2; do
  for v in 0 1; do
    echo "##### rep $rep DSPTOOLBOX_AMD_WELCH_LONG_3PERCU=$v"
    DSPTOOLBOX_AMD_WELCH_LONG_3PERCU=$v timeout -k 10 300 python3 tools/time_welch_sizes.py 16384 32768 65536 2>&1 | tail -8
  done
done
echo "##### parity: FIR tests under FIR_3PERCU=1 / 4, long-window tests under WELCH_LONG_3PERCU=1"
DSPTOOLBOX_AMD_FIR_3PERCU=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "fir and not switches" 2>&1 | tail -3
DSPTOOLBOX_AMD_FIR_3PERCU=4 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "fir and not switches" 2>&1 | tail -3
DSPTOOLBOX_AMD_WELCH_LONG_3PERCU=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "long and not switches" 2>&1 | tail -3
echo done
