#!/bin/bash
# sweep of k_seed_qf's geometry on one box: bash profiles/probes/seed_sweep.sh <tag> "<name:ENV=..,ENV=..> ..." [bench args]
TAG=${1:-x}; VARS=$2; shift; shift
OUT=gpurun_out/seed_sweep_$TAG.txt
ARGS="--no-cpu-baseline --no-secondary --steps 8 --warmup 3 $@"
pick='import sys, json
d = json.loads(sys.stdin.read()); c = d["counters_per_launch"]
print(sys.argv[1], "M reads/s", d["value"], "k_seed alone", d["kernels_ms_one_batch_in_flight"].get("k_seed"), "in flight", d["kernels_ms"].get("k_seed"),
      "wave-trips", sum(c[k] for k in c if k.startswith("seedq_trips")), "slots", sum(c[k] for k in c if k.startswith("seedq_slots")), "idle/phases", c.get("seedq_phases"), "reruns", c.get("reruns_scan_total"))'
for v in $VARS; do
  name=${v%%:*}; envs=${v#*:}
  env $(echo $envs | tr ',' ' ') python bench.py $ARGS 2> gpurun_out/seed_sweep_${TAG}_$name.err | python -c "$pick" "$name[$envs]" >> $OUT
done
cat $OUT
