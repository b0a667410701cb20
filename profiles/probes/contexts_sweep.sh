#!/bin/bash
# More contexts with one stream each (so that every context has a hardware queue of its own): does the rate follow the number of batches in flight?
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
OUT=gpurun_out/contexts_$TAG.txt; : > $OUT
run() { python3 bench.py --no-secondary --no-cpu-baseline "${@:2}" 2> gpurun_out/contexts_$TAG.err | python3 -c "import sys, json; d = json.loads(sys.stdin.read()); print('$1', d['value'], {k: v for k, v in d['kernels_ms'].items() if k in ('k_pair', 'k_report', 'k_seed')})" >> $OUT; }
run two_streams_12 "$@" && DG_ONE_STREAM=1 run one_stream_12 "$@" && DG_ONE_STREAM=1 run one_stream_15 --inflight 15 "$@" && DG_ONE_STREAM=1 GPU_MAX_HW_QUEUES=24 run one_stream_20_q24 --inflight 20 "$@" && DG_ONE_STREAM=1 run one_stream_8 --inflight 8 "$@"
cat $OUT
