#!/bin/bash
# The rate over time of a long run (959 M reads/s over 0.4 s, 862 over 139 s on the same box -- when does it drop, and what changes then?):
# per-1000-batch rates from bench.py (DART_BENCH_RATE_LOG) beside rocm-smi's shader / memory / fabric clocks, temperatures (edge, junction, memory), power, sampled every second.
#   bash profiles/probes/sustained.sh <tag> <steps> [bench args]
TAG=$1; STEPS=$2; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
OUT=gpurun_out/sustained_$TAG
: > $OUT.smi
( while true; do echo "t=$(date +%s.%N)" >> $OUT.smi; /opt/rocm/bin/rocm-smi --showclocks --showpower --showtemp --showuse 2>&1 | grep -E "sclk|mclk|fclk|socclk|Power|GPU use|Temperature" >> $OUT.smi; sleep 1; done ) &
SAMPLER=$!
DART_BENCH_RATE_LOG=$OUT.rates DART_BENCH_RATE_EVERY=1000 python3 bench.py --no-secondary --no-cpu-baseline --steps $STEPS --warmup 3 "$@" > $OUT.json 2> $OUT.err
RC=$?
kill $SAMPLER
python3 - $OUT <<'PY'
import sys, re, json
out = sys.argv[1]
smi = []          # (t, dict)
cur = None
for l in open(out + ".smi"):
    if l.startswith("t="):
        cur = {}; smi.append((float(l[2:]), cur)); continue
    m = re.search(r"(sclk|mclk|fclk|socclk)\s+clock level:?\s*\S*\s*\((\d+)Mhz\)", l)
    if m and cur is not None: cur[m.group(1)] = int(m.group(2))
    m = re.search(r"Power \(W\):\s*([\d.]+)", l)
    if m and cur is not None: cur["W"] = float(m.group(1))
    m = re.search(r"Temperature \(Sensor (\w+)\) \(C\):\s*([\d.]+)", l)
    if m and cur is not None: cur["T_" + m.group(1)] = float(m.group(2))
rates = [(float(a[0]), int(a[2]), float(a[4])) for a in (l.split() for l in open(out + ".rates"))]
print("time_s items rate_M_reads_per_s | nearest rocm-smi sample")
t0 = rates[0][0] if rates else 0
for t, n, r in rates:
    near = min(smi, key=lambda x: abs(x[0] - t))[1] if smi else {}
    print("%7.1f %7d %8.1f | %s" % (t - t0, n, r, " ".join("%s=%s" % kv for kv in sorted(near.items()))))
d = json.load(open(out + ".json"))
print("value over the whole run", d["value"], "M reads/s,", d["steps"], "steps")
PY
exit $RC
