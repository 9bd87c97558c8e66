#!/bin/bash
# gpurun with retries while no slot / box is free (exit code 3: nothing charged).  usage: tools/gpurun_retry.sh <timeout_s> '<command>'
T="$1"; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  [ "$rc" != 3 ] && exit $rc
  sleep 45
done
exit 3
