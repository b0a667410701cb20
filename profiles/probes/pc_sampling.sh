#!/bin/bash
# Where do the waves' program counters sit while twelve batches are in flight?  rocprofv3's PC sampling (beta) on a build with line tables.
#   bash profiles/probes/pc_sampling.sh <tag> <method: host_trap|stochastic> <unit> <interval> [bench args]
TAG=$1; METHOD=$2; UNIT=$3; INTERVAL=$4; shift; shift; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
export TMPDIR=/tmp
LIB=$ROOT/dart_amd/libdartgpu_lines.so
[ -f $LIB ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -gline-tables-only -o $LIB dart_amd/csrc/dg_api.hip 2> /dev/null || exit 1
python3 bench.py --steps 1 --warmup 0 --batches 1 --inflight 1 --no-cpu-baseline --no-secondary "$@" > /dev/null 2> gpurun_out/pcs_$TAG.prep.err || true    # index into the cache
OUT=/tmp/pcs_$TAG; rm -rf $OUT
export DARTGPU_LIB=$LIB
rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit $UNIT --pc-sampling-method $METHOD --pc-sampling-interval $INTERVAL --kernel-trace --output-format csv -d $OUT -- python3 bench.py --steps 12 --warmup 1 --batches 10 --no-cpu-baseline --no-secondary "$@" > gpurun_out/pcs_$TAG.bench.json 2> gpurun_out/pcs_$TAG.err
echo "rocprofv3 exit $?"
find $OUT -type f | xargs ls -la > gpurun_out/pcs_$TAG.files.txt
for f in $(find $OUT -name "*pc_sampling*.csv"); do head -5 $f > gpurun_out/pcs_$TAG.head.txt; done
python3 profiles/probes/pc_sampling_hist.py $OUT > gpurun_out/pcs_$TAG.hist.txt 2> gpurun_out/pcs_$TAG.hist.err
tail -5 gpurun_out/pcs_$TAG.err; cat gpurun_out/pcs_$TAG.files.txt; cat gpurun_out/pcs_$TAG.head.txt; head -50 gpurun_out/pcs_$TAG.hist.txt
