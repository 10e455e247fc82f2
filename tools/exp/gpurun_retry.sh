#!/bin/bash
# dev helper: submit a gpurun call, retrying while the pool has no free box (exit status 3: nothing charged)
#   tools/exp/gpurun_retry.sh <timeout-seconds> '<command>'
for attempt in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  echo "[retry] attempt $attempt: no free box, waiting 150 s"
  sleep 150
done
exit 3
