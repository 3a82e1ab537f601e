#!/bin/bash
# gpurun with retries while every slot is busy (exit code 3): tools/gpurun_retry.sh <timeout-s> '<command>'
T=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"; rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 60
done
exit 3
