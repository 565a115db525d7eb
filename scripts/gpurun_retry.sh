#!/bin/bash
# gpurun with retries on "no box / slot free" (rc 3, nothing charged):  bash scripts/gpurun_retry.sh <timeout-s> '<command>'
T="$1"; shift
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3
