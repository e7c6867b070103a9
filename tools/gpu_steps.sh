#!/bin/bash
# Run a list of GPU steps one after the other on the box gpurun gives us: every step under its own timeout, its output under
# gpurun_out/<tag>/, and NO further step once one was killed at its limit (a hung kernel must not be followed by more GPU work).
#   tools/gpu_steps.sh <tag> "<name>|<seconds>|<command>" ...
tag=$1; shift
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
for step in "$@"; do
    name=${step%%|*}; rest=${step#*|}; secs=${rest%%|*}; cmd=${rest#*|}
    echo "=== $name (limit ${secs}s): $cmd" | tee -a "$out/steps.log"
    t0=$(date +%s)
    timeout -k 10 "$secs" bash -c "$cmd" > "$out/$name.out" 2> "$out/$name.err"
    rc=$?
    echo "=== $name: rc $rc, $(( $(date +%s) - t0 )) s" | tee -a "$out/steps.log"
    tail -n 5 "$out/$name.out" | cut -c1-400
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then
        echo "=== $name was killed at its limit: no further GPU step" | tee -a "$out/steps.log"
        exit 1
    fi
done
exit 0
