#!/bin/bash
# A/B of two builds of the library on one box, alternating: bash profiles/probes/lib_ab.sh <tag> <other.so> [bench args]
TAG=${1:-x}; OTHER=$2; shift; shift
OUT=gpurun_out/lib_ab_$TAG.txt
ARGS="--no-cpu-baseline --steps 20 --warmup 3 $@"
pick='import sys, json
d = json.loads(sys.stdin.read())
k = d["kernels_ms_one_batch_in_flight"]; kf = d["kernels_ms"]
print(sys.argv[1], "M reads/s", d["value"], d.get("value_repeats"), "resident", d.get("value_device_resident"), "ascii", d.get("value_ascii_input"), "full", d.get("value_full_records"))
print("   alone:", {a: round(b, 3) for a, b in k.items()})
print("   in flight:", {a: round(b, 3) for a, b in kf.items()})'
for rep in 1 2; do
  DARTGPU_LIB=$PWD/$OTHER python bench.py $ARGS 2> gpurun_out/lib_ab_${TAG}_other.err | python -c "$pick" "other[$OTHER]" >> $OUT
  python bench.py $ARGS 2> gpurun_out/lib_ab_${TAG}_new.err | python -c "$pick" "new" >> $OUT
done
cat $OUT
