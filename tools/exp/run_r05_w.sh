#!/bin/bash
# round 5: r05c profiles again (the bench line now reads compiler and kernel fingerprints from lib/build_info.json instead of
# starting hipcc / c++filt from a process that has initialised the GPU), the switch tests, the default line untraced
set -u
export TMPDIR=/tmp
echo "##### switch tests"
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "switches" 2>&1 | tail -4 || exit 1
echo "##### profiles (r05c: fir_bank, welch_h1)"
bash tools/prof_all.sh r05c "fir_bank welch_h1" > gpurun_out/r05_w_prof.log 2>&1
tail -3 gpurun_out/r05_w_prof.log
for W in welch_h1 fir_bank; do cp profiles/r05c_${W}_rocprofv3_summary.txt gpurun_out/r05c_${W}_summary.txt; done
echo "##### profile tests on this box"
timeout -k 10 300 python3 -m pytest tests/test_profiles.py -q 2>&1 | tail -5
echo "##### default line, untraced (driver's command)"
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r05_w_line.json 2> gpurun_out/r05_w_line.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_w_line.json').read().strip().splitlines()[-1])
r=d['roofline']; s=d['steady_state']
print('ms',d['ms_per_step'],'frac',r['frac'],'current',r.get('traffic_kernel_current'),'steady',s['roofline_frac'], 'wall', d.get('workloads_wall_s'))
for k,v in d.get('workloads',{}).items(): print(k, v.get('ms_per_step'), v.get('frac'), v.get('traffic_ratio'), v.get('traffic_kernel_current'))
PY
ls -la gpurun_out/.graft_exec_refused 2>/dev/null
echo done
