#!/bin/bash
# round 5: the round's profiles (PMC passes + kernel traces of the five bench workloads), a two-rank rehearsal of
# bench.py on one GPU (host broadcast: RCCL refuses two ranks on a device), kernel resources
set -u
export TMPDIR=/tmp
cd /tmp && cd - > /dev/null
echo "##### resident spectrogram to host"; timeout -k 10 200 python3 tools/time_api_resident.py 2>&1 | grep get_spectrogram
tools/prof_all.sh r05 "welch_h1 welch_h1_1024 fir_bank csm deconv" 2>&1 | tail -8
for W in welch_h1 fir_bank deconv; do
  echo "##### two ranks on one GPU, host broadcast: $W"
  BENCH_BCAST=host timeout -k 10 300 python3 bench.py --gpus 2 --workload $W --steps 20 --warmup 5 --no-cpu-baseline --steady-steps 200 2>&1 | tail -3 | cut -c1-700
done
cp dsptoolbox_amd/lib/kernel_resources.json gpurun_out/r05_kernel_resources.json
echo done
