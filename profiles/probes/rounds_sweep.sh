#!/bin/bash
# seeding rounds: rounds x max Occ steps per search x resident blocks per CU -> pipelined throughput and k_seed stage time alone
CFGS=${ROUND_CFGS:-0,4,16 6,4,16 6,4,3 6,4,2 6,2,3 6,8,3 3,4,3}
for cfg in $CFGS; do
  IFS=, read r s b <<< "$cfg"
  DG_SEED_ROUNDS=$r DG_SEED_ROUND_STEPS=$s DG_SEED_ROUND_BPC=$b python bench.py --no-cpu-baseline "$@" 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('rounds $r steps $s bpc $b:', d['value'], 'M reads/s', d['ms_per_step'], 'ms/step; seed stage alone', d['kernels_ms_one_batch_in_flight']['k_seed'], 'leftover wave trips', d['counters_per_launch']['k_seed_wave_trips_sum'])
"
done
