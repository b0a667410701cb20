# round 5, call as: after d_gap_small.  (1) k_report by cost class and phase again (diagnostic build -DDG_PROFILE_CLASSES of the shipped sources, as r05_ao.sh);
# (2) which form computes the nw_alignment cells: a probe build whose one-lane d_nw adds its cells to the re-seeding window counter and whose d_pair_nw adds them to the
# re-seeding call counter (profiles/probes/dyn/libdartgpu_nwsplit.so), beside the product build on the same single batch -- the differences are the cells of each form
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ONE="--no-secondary --no-cpu-baseline --sustained-s 0 --human-like-budget 0 --inflight 1 --batches 1 --steps 1 --warmup 1"
for w in cfg5 human; do
  case $w in cfg5) A="--rlen 151 --spliced 0.3 --introns 20000";; human) A="--genome-model human";; esac
  DARTGPU_LIB=$GRAFT_REPO_ROOT/profiles/probes/dyn/libdartgpu_class_profile.so timeout -k 10 300 python bench.py $A $ONE > gpurun_out/r05_as_classes_$w.json 2> gpurun_out/r05_as_classes_$w.err || exit 1
  grep -E "^\[class" gpurun_out/r05_as_classes_$w.err | tail -26 > gpurun_out/r05_as_k_report_classes_$w.txt; cut -c1-220 gpurun_out/r05_as_k_report_classes_$w.txt
  timeout -k 10 300 python bench.py $A $ONE > gpurun_out/r05_as_product_$w.json 2> gpurun_out/r05_as_product_$w.err || exit 1
  DARTGPU_LIB=$GRAFT_REPO_ROOT/profiles/probes/dyn/libdartgpu_nwsplit.so timeout -k 10 300 python bench.py $A $ONE > gpurun_out/r05_as_nwsplit_$w.json 2> gpurun_out/r05_as_nwsplit_$w.err || exit 1
done
python - <<'PY'
import json
for w in ("cfg5","human"):
    a=json.loads(open("gpurun_out/r05_as_product_%s.json"%w).read().strip().splitlines()[-1])["counters_per_launch"]
    b=json.loads(open("gpurun_out/r05_as_nwsplit_%s.json"%w).read().strip().splitlines()[-1])["counters_per_launch"]
    lane=b["reseed_window"]-a["reseed_window"]; small=b["reseed_calls"]-a["reseed_calls"]
    print(w, "nw_calls", a["nw_calls"], "cells: all", a["nw_cells"], "| one-lane d_nw", lane, "| d_pair_nw (<= 24 x 24, lane-parallel)", small, "| wave-wide forms", b["nw_cells"], "| check", lane+small+b["nw_cells"]-a["nw_cells"])
PY
