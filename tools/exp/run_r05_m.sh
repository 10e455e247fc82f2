#!/bin/bash
# round 5: float64 median matrix route, then every workload re-profiled so that the summaries carry kernel fingerprints
set -u
export TMPDIR=/tmp
echo "##### median matrix in float64 + the short-estimate tests"
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "short_estimate or csm_median or api_hold" 2>&1 | tail -5 || exit 1
echo "##### profiles"
bash tools/prof_all.sh r05 > gpurun_out/r05_m_prof.log 2>&1
tail -3 gpurun_out/r05_m_prof.log
for W in welch_h1 welch_h1_1024 fir_bank csm deconv; do cp profiles/r05_${W}_rocprofv3_summary.txt gpurun_out/r05m_${W}_summary.txt; done
echo "##### fingerprints against the library on this box"
timeout -k 10 300 python3 -m pytest tests/test_profiles.py -q 2>&1 | tail -5
echo "##### default line"
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r05_m_line.json 2> gpurun_out/r05_m_line.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_m_line.json').read().strip().splitlines()[-1])
r=d['roofline']
print('frac',r['frac'],'traffic',r.get('traffic'),'current',r.get('traffic_kernel_current'))
for k,v in d.get('workloads',{}).items(): print(k, v.get('frac'), v.get('traffic_ratio'))
PY
echo done
