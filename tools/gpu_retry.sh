#!/bin/bash
# usage: tmp_libs/gpu_retry.sh <timeout> <script>   -- retries while the pod has no free GPU slot (exit 3)
for i in $(seq 20); do
  /usr/local/graft/bin/gpurun --timeout $1 -- "bash $2"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3
