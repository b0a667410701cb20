#!/bin/bash
# per-kernel times (one batch in flight) of a bench run: kiso.sh <label> <bench args...>
lab=$1; shift
python bench.py --genome chr20 --no-cpu-baseline --steps 8 --warmup 4 "$@" 2> gpurun_out/kiso_$lab.err | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$lab', 'ms/step', d['ms_per_step'], 'iso', {k: round(v, 3) for k, v in d['kernels_ms_one_batch_in_flight'].items()})
print('   counters', d['counters_per_launch'])
"
