#!/bin/bash
# usage: gpu_retry.sh <timeout> <command>   -- retries while the pod has no free GPU slot (exit code 3)
for i in $(seq 1 20); do
  gpurun --timeout "$1" -- "$2"; rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 60
done
exit 3
