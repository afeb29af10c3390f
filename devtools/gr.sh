#!/bin/bash
# gpurun with retries while the pod's GPU slots are busy (status=transient: nothing charged). Usage: devtools/gr.sh <timeout> '<command>' <logfile>
T=$1; CMD=$2; LOG=$3
for i in $(seq 1 40); do
    /usr/local/graft/bin/gpurun --timeout $T -- "$CMD" > $LOG 2>&1
    rc=$?
    if grep -q "status=transient" $LOG; then sleep 60; continue; fi
    exit $rc
done
exit 3
