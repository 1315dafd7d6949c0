#!/bin/bash
# gpurun with waiting for a free slot: retries ONLY when gpurun reports "no box or slot free" (exit code 3: nothing ran, nothing
# charged); any other outcome -- including a failed or killed command -- is returned at once.
#   tools/debug/gpurun_wait.sh <timeout_s> '<command>'
T=$1; shift
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 150
done
exit 3
