#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun): kernel trace of the DEFAULT bench command first, PMC passes separately.
#   bash profiles/run_profile.sh <tag>
set -e
TAG=${1:-r01}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1 || true   # builds and caches the index outside the profiler
# the default command (24 steps, 8 warm-up, 4 batches in flight), minus the CPU leg
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py --no-cpu-baseline > $OUT/bench_trace.json 2> $OUT/bench_trace.err
# counters: one batch in flight (the profiler serialises dispatches anyway), separate passes
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python bench.py --steps 2 --warmup 0 --inflight 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_tcc -- python bench.py --steps 2 --warmup 0 --inflight 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_tcc.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc_sq -- python bench.py --steps 2 --warmup 0 --inflight 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --pmc SQ_INSTS SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_IFETCH SQC_ICACHE_BUSY_CYCLES --output-format csv -d $OUT/pmc_inst -- python bench.py --steps 2 --warmup 0 --inflight 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_inst.err
python profiles/summarize.py $OUT > $OUT/summary.txt
python profiles/make_traffic.py $OUT $TAG > $OUT/traffic.txt
cat $OUT/summary.txt
