#!/bin/bash
# gpurun with retries while no slot / box is free (exit code 3): tools/gpurun_retry.sh <timeout_s> <command...>
T=$1; shift
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"; rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 60
done
exit 3
