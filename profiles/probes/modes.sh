#!/bin/bash
# The same command in fresh processes, several times: is the rate a property of the process (stream -> hardware queue assignment, memory placement)?
#   bash profiles/probes/modes.sh <tag> <runs> [ENV=VALUE ...] -- [bench args]
TAG=$1; N=$2; shift; shift
ENVS=(); while [ "$1" != "--" ] && [ $# -gt 0 ]; do ENVS+=("$1"); shift; done; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
OUT=gpurun_out/modes_$TAG.txt; : > $OUT
for i in $(seq 1 $N); do
  env "${ENVS[@]}" $MODES_WRAP python3 bench.py --no-secondary --no-cpu-baseline --repeats 2 "$@" 2> gpurun_out/modes_$TAG.err | python3 -c "import sys, json; d = json.loads(sys.stdin.read()); print('$TAG run $i', d['value_repeats'], {k: v for k, v in d['kernels_ms'].items() if k in ('k_pair', 'k_seed')})" >> $OUT || exit 1
done
cat $OUT
