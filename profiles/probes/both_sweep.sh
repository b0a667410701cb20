#!/bin/bash
# k_seed: run a mode's code in a trip only when at least DG_SEED_BOTH lanes wait in it (0 = the shipped policy: cheap modes always, the fuller heavy mode)
for t in ${BOTH_T:-0 8 16 24 32}; do
  DG_SEED_BOTH=$t python bench.py --no-cpu-baseline "$@" 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('both_thr $t:', d['value'], 'M reads/s', d['ms_per_step'], 'ms/step; k_seed alone', d['kernels_ms_one_batch_in_flight']['k_seed'], 'wave trips', d['counters_per_launch']['k_seed_wave_trips_sum'])
"
done
