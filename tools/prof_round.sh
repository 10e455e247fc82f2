#!/bin/bash
# One round's measured evidence in one GPU call: tools/prof_round.sh <tag>
#   profiles of the five bench workloads (tools/prof_all.sh), the per-rank shard prediction of an 8-GPU
#   strong-scaling job, the window-length sweeps (Welch / STFT incl. the four-step lengths), the API end-to-end times.
set -u
TAG=${1:-r04}
tools/prof_all.sh $TAG "${2:-welch_h1 welch_h1_1024 fir_bank csm deconv}"
: > gpurun_out/${TAG}_shard_prediction.jsonl
for W in welch_h1 welch_h1_1024 fir_bank csm deconv; do
  python3 bench.py --workload $W --steps 100 --warmup 10 --predict-ranks 8 --no-cpu-baseline 2>> gpurun_out/${TAG}_shard_prediction.err | grep '^{' >> gpurun_out/${TAG}_shard_prediction.jsonl
  echo "shard prediction $W done"
done
python3 tools/time_welch_sizes.py 32 128 256 1024 2048 4096 8192 16384 32768 65536 262144 > gpurun_out/${TAG}_welch_sizes.log 2>&1
echo "welch sizes done"
python3 tools/time_stft_sizes.py 256 1024 4096 8192 16384 32768 65536 > gpurun_out/${TAG}_stft_sizes.log 2>&1
echo "stft sizes done"
python3 tools/time_api_e2e.py > gpurun_out/${TAG}_api_e2e.log 2>&1
echo "api e2e done"
