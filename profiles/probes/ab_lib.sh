#!/bin/bash
# A/B of two builds of the library on one box: bash profiles/probes/ab_lib.sh <tag> <other .so> [bench args]
# (the default build first and last, the other one in between; --no-secondary --no-cpu-baseline lines, value only)
TAG=$1; OTHER=$2; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
OUT=gpurun_out/ab_$TAG.txt; : > $OUT
run() { python3 bench.py --no-secondary --no-cpu-baseline --repeats 1 "${@:2}" 2> /dev/null | python3 -c "import sys, json; d = json.loads(sys.stdin.read()); print('$1', d['value'], d['value_repeats'], {k: v for k, v in d['kernels_ms'].items() if k in ('k_pair', 'k_report', 'k_seed')}, 'alone', {k: v for k, v in d['kernels_ms_one_batch_in_flight'].items() if k in ('k_pair', 'k_report', 'k_seed')})" >> $OUT; }
run default "$@" && DARTGPU_LIB=$ROOT/$OTHER run other "$@" && run default "$@" && DARTGPU_LIB=$ROOT/$OTHER run other "$@"
cat $OUT
