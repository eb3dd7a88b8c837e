#!/bin/bash
# gpurun with patience: retries ONLY when no GPU slot / box was free (exit code 3: nothing ran, nothing was charged), every
# two minutes, at most 15 times.  Any other outcome -- success or a failure of the command itself -- is returned at once.
# usage: tools/gpurun_wait.sh <gpurun args...>
for attempt in $(seq 1 15); do
  /usr/local/graft/bin/gpurun "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 120
done
exit 3
