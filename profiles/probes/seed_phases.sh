#!/bin/bash
# Where the cycles of a k_seed_qf wave go: a probe build of the library (-DDG_SQF_PROF: s_memtime stamps between the parts of a trip, waits forced
# at the stamps) prints the sums of a few waves per launch.   bash profiles/probes/seed_phases.sh <tag> [bench args]
TAG=${1:-x}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
LIB=$ROOT/dart_amd/libdartgpu_prof.so
[ -f $LIB ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -DDG_SQF_PROF -o $LIB dart_amd/csrc/dg_api.hip 2> /dev/null
for fl in 1 12; do
  DARTGPU_LIB=$LIB python3 bench.py --steps 2 --warmup 1 --batches 4 --inflight $fl --no-cpu-baseline --no-secondary "$@" > gpurun_out/seed_phases_${TAG}_inflight$fl.txt 2> gpurun_out/seed_phases_${TAG}_inflight$fl.err
  python3 - gpurun_out/seed_phases_${TAG}_inflight$fl.txt <<'PY'
import sys, re, collections
rows = [l for l in open(sys.argv[1]) if l.startswith("sqf wg")]
tot = collections.OrderedDict(); n = 0; cyc = 0; trips = 0; looks = 0
for l in rows:
    m = re.search(r"total (\d+) trips (\d+) looks (\d+) \| (.*)", l)
    cyc += int(m.group(1)); trips += int(m.group(2)); looks += int(m.group(3)); n += 1
    it = m.group(4).split()
    for k in range(0, len(it), 2): tot[it[k]] = tot.get(it[k], 0) + int(it[k + 1])
print(sys.argv[1], ": %d waves, mean cycles %.0f, trips %.1f, looks %.1f" % (n, cyc / max(n, 1), trips / max(n, 1), looks / max(n, 1)))
s = sum(tot.values())
print("  " + "  ".join("%s %.1f%%" % (k, 100.0 * v / max(s, 1)) for k, v in tot.items()), " (stamped %.1f%% of total)" % (100.0 * s / max(cyc, 1)))
print("  per trip: " + "  ".join("%s %.0f" % (k, v / max(trips, 1)) for k, v in tot.items()))
PY
done
