#!/bin/bash
# stand-alone seeding time and host-to-host rate per bail-out threshold
for m in planted human; do for b in 64 48 32 24 16; do
  DG_SEED_BAIL_TRIPS=$b python bench.py --genome-model $m --no-cpu-baseline --no-secondary --steps 10 --warmup 3 2> gpurun_out/r4l/$m_$b.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['kernels_ms_one_batch_in_flight']; c=d['counters_per_launch']
print('$m bail $b: value', d['value'], 'k_seed alone', round(k['k_seed'],3), 'in flight', round(d['kernels_ms']['k_seed'],2), 'wave_ms qf', round(c['wave_ticks_k_seed_qf']*1e-5), 'heavy', round(c['wave_ticks_k_seed_heavy']*1e-5), 'steps_exec', c['steps_executed'])"
done; done
