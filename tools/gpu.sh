#!/bin/bash
# tools/gpu.sh TIMEOUT_SECONDS 'command' -- one gpurun call; waits and asks again ONLY while no GPU slot / box is free (exit code 3:
# nothing ran, nothing was charged).  Any other outcome, success or failure of the command itself, is returned as it is.
t=$1; shift
for try in $(seq 1 30); do
    /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
    rc=$?
    if [ $rc -ne 3 ]; then exit $rc; fi
    echo "[gpu.sh] no slot (try $try), waiting" >&2
    sleep 100
done
exit 3
