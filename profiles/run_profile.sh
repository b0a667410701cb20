#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun): kernel trace of the DEFAULT bench command first, PMC passes separately.
#   bash profiles/run_profile.sh <tag> [extra bench.py arguments, e.g. --genome chr20]
# Writes gpurun_out/prof_<tag>/{summary.txt,traffic.txt,bench_trace.json,...}; copy what is to be kept into profiles/<round>/.
TAG=${1:-r02}; shift
EXTRA="$@"
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $ROOT
python3 bench.py --steps 1 --warmup 0 --batches 1 --inflight 1 --no-cpu-baseline --no-secondary $EXTRA > /dev/null 2> $OUT/prep.err || true   # builds and caches the index outside the profiler
# the default command (ten distinct batches per step, the default number in flight), fewer steps, minus the CPU leg
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary $EXTRA > $OUT/bench_trace.json 2> $OUT/bench_trace.err
echo "trace done"
# counters: one batch in flight (the profiler serialises dispatches anyway), separate passes (PROFILE_PASSES="fetch tcc" runs a subset: a call on the GPU box is limited to 20 minutes)
PMC="--steps 1 --warmup 0 --batches 2 --inflight 1 --no-cpu-baseline --no-secondary $EXTRA"
PASSES=${PROFILE_PASSES:-fetch tcc sq inst}
[[ " $PASSES " == *" fetch "* ]] && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $PMC > /dev/null 2> $OUT/pmc_fetch.err
echo "pmc fetch done"
[[ " $PASSES " == *" tcc "* ]] && rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_tcc -- python3 bench.py $PMC > /dev/null 2> $OUT/pmc_tcc.err
echo "pmc tcc done"
[[ " $PASSES " == *" sq "* ]] && rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc_sq -- python3 bench.py $PMC > /dev/null 2> $OUT/pmc_sq.err
echo "pmc sq done"
[[ " $PASSES " == *" inst "* ]] && rocprofv3 --pmc SQ_INSTS SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_IFETCH SQC_ICACHE_BUSY_CYCLES --output-format csv -d $OUT/pmc_inst -- python3 bench.py $PMC > /dev/null 2> $OUT/pmc_inst.err
echo "pmc inst done"
python3 profiles/summarize.py $OUT > $OUT/summary.txt
python3 profiles/make_traffic.py $OUT $TAG > $OUT/traffic.txt
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_tcc $OUT/pmc_sq $OUT/pmc_inst        # (the raw trees are hundreds of MB)
cat $OUT/summary.txt | head -60
