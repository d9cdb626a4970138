#!/bin/bash
# gpurun with patience: exit code 3 means "no box or slot free right now, nothing charged" — wait and ask again (at most 12 times).
# Any other outcome (the command ran, was refused, timed out) is returned as it is: a GPU step that ran is never repeated here.
#   tools/gpu_retry.sh TIMEOUT 'command'
t=$1; shift
for i in $(seq 1 12); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 100
done
exit 3
