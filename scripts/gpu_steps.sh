#!/bin/bash
# Runs GPU steps one after the other on the gpurun box; a step that TIMES OUT or is KILLED ends the call (no further GPU step after
# a hang), an ordinary failure (assertion, non-zero exit) is logged and the next step still runs.
#   scripts/gpu_steps.sh "<seconds> <log name> <command...>" ...
mkdir -p gpurun_out
for spec in "$@"; do
    set -- $spec
    secs=$1; name=$2; shift 2
    echo "=== [$name] timeout ${secs}s: $*" | tee -a gpurun_out/steps.log
    start=$(date +%s)
    timeout -k 10 "$secs" "$@" > "gpurun_out/$name.log" 2>&1
    rc=$?
    echo "=== [$name] rc=$rc in $(( $(date +%s) - start )) s" | tee -a gpurun_out/steps.log
    tail -n 6 "gpurun_out/$name.log"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "=== [$name] timed out / killed: stopping" | tee -a gpurun_out/steps.log; exit $rc; fi
done
exit 0
