#!/bin/bash
# Shader / memory clocks and power while the timed region runs (is the rate limited by a throttled clock?): rocm-smi sampled every 0.5 s beside the bench.
#   bash profiles/probes/clocks_during_bench.sh <tag> [bench args]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
OUT=gpurun_out/clocks_$TAG.txt; : > $OUT
( while true; do echo "t=$(date +%s.%N)" >> $OUT; /opt/rocm/bin/rocm-smi --showclocks --showpower --showuse 2>&1 | grep -E "sclk|mclk|fclk|Power|GPU use" >> $OUT; sleep 0.5; done ) &
SAMPLER=$!
python3 bench.py --no-secondary --no-cpu-baseline --steps 200 --warmup 3 "$@" > gpurun_out/clocks_$TAG.json 2> gpurun_out/clocks_$TAG.err
RC=$?
kill $SAMPLER
python3 - $OUT <<'PY'
import sys, re, collections
vals = collections.defaultdict(list)
for l in open(sys.argv[1]):
    m = re.search(r"(sclk|mclk|fclk)\s+clock level:?\s*\S*\s*\((\d+)Mhz\)", l)
    if m: vals[m.group(1)].append(int(m.group(2)))
    m = re.search(r"Power \(W\):\s*([\d.]+)", l)
    if m: vals["power_W"].append(float(m.group(1)))
    m = re.search(r"GPU use \(%\):\s*(\d+)", l)
    if m: vals["gpu_use"].append(int(m.group(1)))
for k, v in vals.items():
    print(k, "samples", len(v), "min", min(v), "max", max(v), "last 10:", v[-10:])
PY
python3 -c "import json; d = json.load(open('gpurun_out/clocks_$TAG.json')); print('value', d['value'], 'ms_per_step', d['ms_per_step'])"
exit $RC
