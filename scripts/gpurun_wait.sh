#!/bin/bash
# gpurun with patience: exit code 3 means "no box / slot free, nothing ran, nothing charged" — wait and ask again (up to ~40 min).
# Any other exit code is the call's own verdict and is returned as is (a command that ran is never repeated).
#   scripts/gpurun_wait.sh <timeout seconds> '<command>'
for attempt in $(seq 1 20); do
    /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
    rc=$?
    [ $rc -ne 3 ] && exit $rc
    sleep 120
done
exit 3
