#!/bin/bash
# stand-alone seeding time and host-to-host rate per drain-phase bail-out threshold (DG_SEED_DRAIN_BAIL; 0 = the same as everywhere: 64)
for m in planted human; do for b in 0 32 16 8 4; do
  DG_SEED_DRAIN_BAIL=$b python bench.py --genome-model $m --no-cpu-baseline --no-secondary --steps 10 --warmup 3 2> /dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['kernels_ms_one_batch_in_flight']; c=d['counters_per_launch']
print('$m drain bail $b: value', d['value'], 'k_seed alone', round(k['k_seed'],3), 'in flight', round(d['kernels_ms']['k_seed'],2), 'wave_ms qf', round(c['wave_ticks_k_seed_qf']*1e-5), 'heavy', round(c['wave_ticks_k_seed_heavy']*1e-5))"
done; done
