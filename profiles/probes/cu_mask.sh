#!/bin/bash
# Is the timed region bound by the compute units?  The same bench line with the process's queues confined to a part of them (HSA_CU_MASK).
#   bash profiles/probes/cu_mask.sh <tag> [bench args]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
OUT=gpurun_out/cu_mask_$TAG.txt; : > $OUT
run() { python3 bench.py --no-secondary --no-cpu-baseline "${@:2}" 2> gpurun_out/cu_mask_$TAG.err | python3 -c "import sys, json; d = json.loads(sys.stdin.read()); print('$1', d['value'], d['value_repeats'], 'device-resident', d.get('value_device_resident'), {k: v for k, v in d['kernels_ms_one_batch_in_flight'].items() if k in ('k_pair', 'k_report', 'k_seed')})" >> $OUT; }
run all "$@" && HSA_CU_MASK=0:0-191 run cus_0_191 "$@" && HSA_CU_MASK=0:0-127 run cus_0_127 "$@" && HSA_CU_MASK=0:0-63 run cus_0_63 "$@"
cat $OUT
