#!/bin/bash
# gpurun with a wait when no GPU slot / box is free (exit code 3: nothing was charged).  usage: tools/gpu_retry.sh TIMEOUT 'command'
T=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
