#!/bin/bash
# instruction mix of the seeding kernels per variant: bash profiles/probes/seed_pmc.sh <tag> [bench args]   (one PMC pass per variant)
TAG=${1:-x}; shift
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
python3 bench.py --steps 1 --warmup 0 --batches 1 --inflight 1 --no-cpu-baseline --no-secondary "$@" > /dev/null 2> gpurun_out/seed_pmc_$TAG.prep.err || true
for v in ${SEED_PMC_VARIANTS:-phases:DG_SEED_PHASES=1 free:DG_SEED_PHASES=0}; do
  name=${v%%:*}; envs=${v#*:}
  OUT=$ROOT/gpurun_out/seed_pmc_${TAG}_$name
  mkdir -p $OUT
  export $(echo $envs | tr ',' ' ')
  rocprofv3 --pmc SQ_INSTS SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/pmc_inst -- python3 bench.py --steps 1 --warmup 0 --batches 2 --inflight 1 --no-cpu-baseline --no-secondary "$@" > /dev/null 2> $OUT/pmc_inst.err
  python3 profiles/summarize.py $OUT > $OUT/summary.txt
  rm -rf $OUT/pmc_inst
  echo "== $name [$envs]"; grep "k_seed" $OUT/summary.txt
  unset $(echo $envs | tr ',' ' ' | sed 's/=[^ ]*//g')
done
