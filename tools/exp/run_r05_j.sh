#!/bin/bash
# round 5: two-rank rehearsals of bench.py on one GPU (host broadcast), weak and strong scaling, every workload; new tests
set -u
export TMPDIR=/tmp
echo "##### new tests"; timeout -k 10 300 python3 -m pytest tests -m gpu -x -q -k "resident or welch_2048" 2>&1 | tail -3
for S in weak strong; do
  for W in welch_h1 welch_h1_1024 fir_bank csm deconv; do
    echo "##### two ranks, $S scaling: $W"
    BENCH_BCAST=host timeout -k 10 300 python3 bench.py --gpus 2 --scaling $S --workload $W --steps 20 --warmup 5 --no-cpu-baseline --steady-steps 100 > gpurun_out/r05_j_${S}_$W.json 2> gpurun_out/r05_j_${S}_$W.err; echo "rc $?"; cut -c1-260 gpurun_out/r05_j_${S}_$W.json; tail -2 gpurun_out/r05_j_${S}_$W.err | cut -c1-300
  done
done
echo done
