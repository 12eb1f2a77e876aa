#!/bin/bash
# local helper (this container): retry a gpurun call while the pod's GPU slots are busy (exit code 3 = nothing charged) or another call of this repo is still running (2)
# usage: gpurun_retry.sh <timeout-seconds> <command...>
T=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"; rc=$?
  if [ $rc -ne 3 ] && [ $rc -ne 2 ]; then exit $rc; fi
  sleep 90
done
exit 3
