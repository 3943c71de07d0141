#!/bin/bash
# gpurun with retries while the pod's GPU slots are busy: scratch/gpurun_retry.sh <timeout-seconds> <logfile> <command...>
T=$1; LOG=$2; shift 2
for i in $(seq 1 12); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@" > $LOG 2>&1
  if ! grep -q "status=transient" $LOG; then exit 0; fi
  sleep 90
done
