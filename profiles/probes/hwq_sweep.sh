#!/bin/bash
# pipelined throughput against the number of hardware queues the HIP runtime maps streams onto
#   (GPU_MAX_HW_QUEUES, default 4: a context uses two streams, so four batches in flight already share queues)
python bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
for cfg in ${HWQ_CFGS:-"4,4 8,4 8,8 16,8"}; do
  q=${cfg%,*}; k=${cfg#*,}
  GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu-baseline --inflight $k --steps $((k * 6)) --warmup $((k * 2)) 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('hw_queues $q inflight $k:', d['value'], 'M reads/s', d['ms_per_step'], 'ms/step')
"
done
