#!/bin/bash
# Do fewer different kernels per compute unit help?  Contexts confined to parts of the GPU (DG_CU_PARTS / DG_CU_LAYOUT, dg_api.hip make_ctx_objects).
#   bash profiles/probes/cu_parts.sh <tag> [bench args]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
OUT=gpurun_out/cu_parts_$TAG.txt; : > $OUT
run() { python3 bench.py --no-secondary --no-cpu-baseline "${@:2}" 2> gpurun_out/cu_parts_$TAG.err | python3 -c "import sys, json; d = json.loads(sys.stdin.read()); print('$1', d['value'], {k: v for k, v in d['kernels_ms'].items() if k in ('k_pair', 'k_report', 'k_seed')})" >> $OUT; }
run all "$@" && DG_CU_PARTS=2 DG_CU_LAYOUT=0 run parts2_contiguous "$@" && DG_CU_PARTS=2 DG_CU_LAYOUT=1 run parts2_mod8 "$@" && DG_CU_PARTS=4 DG_CU_LAYOUT=0 run parts4_contiguous "$@" && DG_CU_PARTS=4 DG_CU_LAYOUT=1 run parts4_mod8 "$@" && DG_CU_PARTS=8 DG_CU_LAYOUT=1 run parts8_mod8 "$@"
cat $OUT
