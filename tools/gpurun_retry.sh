#!/bin/bash
# gpurun with retries while no box / slot is free (exit code 3: nothing charged): tools/gpurun_retry.sh <timeout s> '<command>'
t=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 75
done
exit 3
