#!/bin/bash
# round 5: k_fir3 with the first partition's tap spectrum requested first, against k_fir<2> (same box, alternating)
set -u
export TMPDIR=/tmp
for rep in 1 2 3 4; do
  for v in 0 1; do
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
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "fir" 2>&1 | tail -3
echo done
