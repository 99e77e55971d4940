#!/bin/bash
# gpurun with patience: exit code 3 = no box / slot free right now (nothing charged) -> wait and ask again, up to 10 times.
# usage: tools/gpurun_retry.sh <timeout seconds> '<command>'
T=$1; shift
for i in 1 2 3 4 5 6 7 8 9 10; do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3
