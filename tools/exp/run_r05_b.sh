#!/bin/bash
# round 5, second box: the persistent deconvolution kernel (A/B), new tests, device-resident API timing, the bench line with preheat
set -u
export TMPDIR=/tmp
echo "##### tests"; timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q -k "deconv or switch or csm_short or api_rest or fir_signal_shorter or device_resident" 2>&1 | tail -15
echo "##### device-resident API"; timeout -k 10 300 python3 tools/time_api_resident.py 2>&1 | tail -20
for rep in 1 2; do
  echo "##### deconv persist (default) rep $rep"; timeout -k 10 200 python3 bench.py --workload deconv --steps 400 --warmup 20 --no-cpu-baseline --steady-steps 4000 || exit 1
  echo "##### deconv one unit per workgroup (k_deconv3q) rep $rep"; DSPTOOLBOX_AMD_DECONV_PERSIST=0 timeout -k 10 200 python3 bench.py --workload deconv --steps 400 --warmup 20 --no-cpu-baseline --steady-steps 4000 || exit 1
done
echo "##### default bench line (driver command)"; timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 || exit 1
echo done
