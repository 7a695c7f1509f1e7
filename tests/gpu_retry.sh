#!/bin/bash
# retry a gpurun command until a slot is free (exit code 3 = transient). usage: gpu_retry.sh <timeout> '<cmd>' <logfile>
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $1 -- "$2" > "$3" 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3
