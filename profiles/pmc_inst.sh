#!/bin/bash
# one PMC pass (instruction counts per kernel) of a short bench run: bash profiles/pmc_inst.sh <tag> [bench.py arguments]
TAG=${1:-x}; shift
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd $ROOT
python3 bench.py --steps 1 --warmup 0 --batches 1 --inflight 1 --no-cpu-baseline --no-secondary "$@" > /dev/null 2> $OUT/prep.err || true
rocprofv3 --pmc SQ_INSTS SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/pmc_inst -- python3 bench.py --steps 1 --warmup 0 --batches 2 --inflight 1 --no-cpu-baseline --no-secondary "$@" > /dev/null 2> $OUT/pmc_inst.err
python3 profiles/summarize.py $OUT > $OUT/summary.txt
rm -rf $OUT/pmc_inst
grep "k_seed\|k_pair\|k_report\|k_chain" $OUT/summary.txt
