# round 5, call ao (after the splice windows and the 16-lane alignment groups): k_report by cost class and phase on the spliced 2x151 shape and the human-like genome (diagnostic build -DDG_PROFILE_CLASSES of the shipped sources,
# loaded through DARTGPU_LIB; shader cycles per 64-candidate chunk, lane 0's clock)
#   phases: 0 setup | 1 candidate+jobs | 2 seed extension | 3 splice check | 4 normal pairs+validity | 5 classify pairs
#           6 wave-wide NW of large pairs | 7 lane NWs of small pairs | 8 assemble | 9 coordinates+CIGAR out | 10 store
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export DARTGPU_LIB=$GRAFT_REPO_ROOT/profiles/probes/dyn/libdartgpu_class_profile.so
timeout -k 10 300 python bench.py --rlen 151 --spliced 0.3 --introns 20000 --no-secondary --no-cpu-baseline --sustained-s 0 --human-like-budget 0 --inflight 1 --batches 1 --steps 1 --warmup 1 > gpurun_out/r05_al_cfg5.json 2> gpurun_out/r05_al_cfg5.err || exit 1
grep -E "^\[class" gpurun_out/r05_al_cfg5.err | tail -26 > gpurun_out/r05_al_k_report_classes_cfg5.txt; cat gpurun_out/r05_al_k_report_classes_cfg5.txt | cut -c1-220
timeout -k 10 300 python bench.py --genome-model human --no-secondary --no-cpu-baseline --sustained-s 0 --inflight 1 --batches 1 --steps 1 --warmup 1 > gpurun_out/r05_al_human.json 2> gpurun_out/r05_al_human.err || exit 1
grep -E "^\[class" gpurun_out/r05_al_human.err | tail -26 > gpurun_out/r05_al_k_report_classes_human.txt; cat gpurun_out/r05_al_k_report_classes_human.txt | cut -c1-220
