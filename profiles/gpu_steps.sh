#!/bin/bash
# runs GPU steps one after the other, logging each under gpurun_out/; stops at the first step that was killed by its timeout
# usage: profiles/gpu_steps.sh TAG "name1|seconds|command" "name2|seconds|command" ...
tag=$1; shift
mkdir -p gpurun_out
for spec in "$@"; do
    name=${spec%%|*}; rest=${spec#*|}; secs=${rest%%|*}; cmd=${rest#*|}
    echo "== $name (limit ${secs}s): $cmd"
    timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/${tag}_${name}.log" 2>&1
    rc=$?
    echo "== $name rc=$rc"; tail -n 6 "gpurun_out/${tag}_${name}.log"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "== $name was killed at its limit: stopping"; exit 1; fi
done
exit 0
