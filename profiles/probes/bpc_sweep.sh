#!/bin/bash
# pipelined throughput against k_report's persistent waves per CU (DG_REPORT_BPC) and k_seed's (DG_SEED_WAVES)
python bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
for cfg in "8 4" "6 4" "5 4" "4 4" "6 6" "4 8" "8 3"; do
  set -- $cfg
  DG_REPORT_BPC=$1 DG_SEED_WAVES=$2 python bench.py --no-cpu-baseline --inflight 4 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('report_bpc $1 seed_waves $2:', d['value'], 'M reads/s', d['ms_per_step'], 'ms/step')
"
done
