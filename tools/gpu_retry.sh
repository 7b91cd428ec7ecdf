#!/bin/bash
# Submit ONE gpurun call, retrying only while no GPU slot/box is free (exit code 3: nothing ran, nothing was charged).
# usage: tools/gpu_retry.sh <timeout_s> '<command>'      (log: /tmp/gpurun_last.log)
t=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $t -- "$@" > /tmp/gpurun_last.log 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then echo "gpurun rc=$rc after $i attempt(s)" >> /tmp/gpurun_last.log; exit $rc; fi
  sleep 45
done
echo "gave up: no slot" >> /tmp/gpurun_last.log
exit 3
