#!/bin/bash
# usage: tools/r03/gpurun_retry.sh TIMEOUT LOG 'command'  -- repeats the gpurun CALL while the pod has no free GPU slot (rc 3: nothing charged)
T=$1; LOG=$2; CMD=$3
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$CMD" > $LOG 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
