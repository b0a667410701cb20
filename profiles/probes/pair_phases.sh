#!/bin/bash
# Where the cycles of a k_pair wave go: a probe build of the library (-DDG_PAIR_PROF: s_memtime stamps between the phases, lane 0 of each wave, a few
# tiles per launch print their sums).  A lane's stamp is taken when the WAVE reaches it, so a phase's figure is the wave's slowest lane.
#   bash profiles/probes/pair_phases.sh <tag> [bench args]
TAG=${1:-x}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
LIB=$ROOT/dart_amd/libdartgpu_pprof.so
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -DDG_PAIR_PROF -o $LIB dart_amd/csrc/dg_api.hip 2> /dev/null || exit 1
for fl in 1 12; do
  DARTGPU_LIB=$LIB python3 bench.py --steps 2 --warmup 1 --batches 4 --inflight $fl --no-cpu-baseline --no-secondary "$@" > gpurun_out/pair_phases_${TAG}_inflight$fl.txt 2> gpurun_out/pair_phases_${TAG}_inflight$fl.err
  python3 - gpurun_out/pair_phases_${TAG}_inflight$fl.txt <<'PY'
import sys, re, collections
rows = [l for l in open(sys.argv[1]) if l.startswith("kpair tile")]
tot = collections.OrderedDict(); n = 0; cyc = 0
for l in rows:
    m = re.search(r"total (\d+) \| (.*)", l)
    cyc += int(m.group(1)); n += 1
    it = m.group(2).split()
    for k in range(0, len(it), 2): tot[it[k]] = tot.get(it[k], 0) + int(it[k + 1])
print(sys.argv[1], ": %d waves, mean s_memtime ticks per wave %.0f" % (n, cyc / max(n, 1)))
s = sum(tot.values())
print("  " + "  ".join("%s %.1f%%" % (k, 100.0 * v / max(s, 1)) for k, v in tot.items()))
print("  ticks per wave: " + "  ".join("%s %.0f" % (k, v / max(n, 1)) for k, v in tot.items()))
PY
done
