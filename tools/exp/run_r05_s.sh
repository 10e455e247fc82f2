#!/bin/bash
# round 5: matrices through the page-locked staging buffer (tests + end-to-end time), and one more sample of the box spread
set -u
export TMPDIR=/tmp
echo "##### csm / das / device tests"
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "csm or das or beamform or device_resident or fused_float64" 2>&1 | tail -4 || exit 1
echo "##### x64_cap_time (csm rows)"
timeout -k 10 600 python3 tools/x64_cap_time.py 2>&1 | grep "^csm" | tail -20
echo "##### headline on this box (driver's command, workload entries off)"
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-workloads --no-cpu-baseline > gpurun_out/r05_s_line.json 2> gpurun_out/r05_s_line.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_s_line.json').read().strip().splitlines()[-1])
r=d['roofline']; s=d.get('steady_state') or {}
print('ms_per_step',d['ms_per_step'],'kernel_avg_ms',r['kernel_avg_ms'],'frac',r['frac'],'steady frac',s.get('roofline_frac'),'steady ms',s.get('ms_per_step'))
PY
echo done
