#!/bin/bash
# throughput against the number of batches in flight: inflight_sweep.sh <bench args...>
for k in 2 4 6 8; do
  python bench.py --no-cpu-baseline --inflight $k "$@" 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('inflight $k:', d['value'], 'M reads/s', d['ms_per_step'], 'ms/step')
"
done
