#!/bin/bash
# A/B of the seeding kernels on one box: phased queue kernel (DG_SEED_PHASES=1) against the free-running one (default), chr20-sized or any bench arguments:
#   bash profiles/probes/seed_ab.sh <tag> [bench args]       -> gpurun_out/seed_ab_<tag>.txt
TAG=${1:-x}; shift
OUT=gpurun_out/seed_ab_$TAG.txt
ARGS="--no-cpu-baseline --no-secondary --steps 8 --warmup 3 $@"
pick='import sys, json
d = json.loads(sys.stdin.read()); c = d["counters_per_launch"]
print(sys.argv[1], "M reads/s", d["value"], "k_seed alone", d["kernels_ms_one_batch_in_flight"].get("k_seed"), "in flight", d["kernels_ms"].get("k_seed"),
      "wave-trips", sum(c[k] for k in c if k.startswith("seedq_trips")), "slots", sum(c[k] for k in c if k.startswith("seedq_slots")), "idle/phases", c.get("seedq_phases"), "reruns", c.get("reruns_scan_total"))'
for rep in 1 2; do
  for v in "phases:DG_SEED_PHASES=1" "free:DG_SEED_PHASES=0" $SEED_AB_EXTRA; do
    name=${v%%:*}; envs=${v#*:}
    env $(echo $envs | tr ',' ' ') python bench.py $ARGS 2> gpurun_out/seed_ab_${TAG}_$name.err | python -c "$pick" "$name[$envs]" >> $OUT
  done
done
cat $OUT
