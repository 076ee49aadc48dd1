#!/bin/bash
# gpurun with patience: retries while every GPU slot of the pod is busy (exit code 3, nothing charged).  usage: tools/gpu.sh <timeout_s> '<command>'
t=$1; shift
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 60
done
exit 3
