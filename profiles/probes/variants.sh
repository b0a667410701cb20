#!/bin/bash
# one bench run per variant on one box: bash profiles/probes/variants.sh <tag> "<name:lib.so|-:ENV=..,ENV=..> ..." [bench args]
TAG=${1:-x}; VARS=$2; shift; shift
OUT=gpurun_out/variants_$TAG.txt
ARGS="--no-cpu-baseline --no-secondary --steps 20 --warmup 3 $@"
pick='import sys, json
d = json.loads(sys.stdin.read())
print(sys.argv[1], "M reads/s", d["value"], d.get("value_repeats"))
print("   in flight:", {a: round(b, 2) for a, b in d["kernels_ms"].items()})'
for v in $VARS; do
  name=${v%%:*}; rest=${v#*:}; lib=${rest%%:*}; envs=${rest#*:}
  [ "$envs" == "$rest" ] && envs=""
  ( [ "$lib" != "-" ] && export DARTGPU_LIB=$PWD/$lib; [ -n "$envs" ] && export $(echo $envs | tr ',' ' '); python bench.py $ARGS 2> gpurun_out/variants_${TAG}_$name.err | python -c "$pick" "$name[$lib $envs]" >> $OUT )
done
cat $OUT
