#!/bin/bash
# extra SQ/SQC counter passes for the seeding kernels (instruction mix, I-cache); run through gpurun after run_profile.sh
set -e
TAG=${1:-r01}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1 || true
rocprofv3 --pmc SQ_INSTS SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_IFETCH --output-format csv -d $OUT/pmc_x1 -- python bench.py --steps 2 --warmup 0 --no-cpu-baseline > /dev/null 2> $OUT/pmc_x1.err
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES SQC_DCACHE_REQ SQC_DCACHE_MISSES --output-format csv -d $OUT/pmc_x2 -- python bench.py --steps 2 --warmup 0 --no-cpu-baseline > /dev/null 2> $OUT/pmc_x2.err
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/pmc_x3 -- python bench.py --steps 2 --warmup 0 --no-cpu-baseline > /dev/null 2> $OUT/pmc_x3.err
python profiles/summarize.py $OUT > $OUT/summary_extra.txt
