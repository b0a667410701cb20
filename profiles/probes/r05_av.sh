# round 5, call av: full-size parity on the round's final kernels (the bench's ten 1 M-pair batches on the GRCh38-sized index and 1 M spliced 2x151 pairs, every record
# against the oracle), then k_report by cost class and phase on the final sources (diagnostic build, as r05_as.sh)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( while true; do sleep 60; date +%s >> gpurun_out/r05_av_heartbeat.txt; done ) &
HB=$!
timeout -k 10 500 python tests/probes/full_size_parity.py both 10 > gpurun_out/r05_av_full_size_parity.txt 2>&1; echo "parity rc=$?"
tail -6 gpurun_out/r05_av_full_size_parity.txt | cut -c1-250
ONE="--no-secondary --no-cpu-baseline --sustained-s 0 --human-like-budget 0 --inflight 1 --batches 1 --steps 1 --warmup 1"
for w in cfg5 human; do
  case $w in cfg5) A="--rlen 151 --spliced 0.3 --introns 20000";; human) A="--genome-model human";; esac
  DARTGPU_LIB=$GRAFT_REPO_ROOT/profiles/probes/dyn/libdartgpu_class_profile.so timeout -k 10 200 python bench.py $A $ONE > gpurun_out/r05_av_classes_$w.json 2> gpurun_out/r05_av_classes_$w.err || break
  grep -E "^\[class" gpurun_out/r05_av_classes_$w.err | tail -13 > gpurun_out/r05_av_k_report_classes_$w.txt; cut -c1-220 gpurun_out/r05_av_k_report_classes_$w.txt
done
kill $HB
