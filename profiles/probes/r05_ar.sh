# round 5, call ar: a build with three instruction-count changes (profiles/probes/dyn/libdartgpu_pack1.so: one truncation per nw_alignment cell instead of three, k_reseed's filter
# bits gathered by v_alignbit, read gaps of <= 24 bases filled by the lane itself without strings: d_gap_small) beside the product build: the whole GPU suite and 30 fuzz rounds
# THROUGH the variant (DARTGPU_LIB), then spliced 2x151 / human-like / planted rates A / B on the same box
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( while true; do sleep 60; date +%s >> gpurun_out/r05_ar_heartbeat.txt; done ) &
HB=$!
V=$GRAFT_REPO_ROOT/profiles/probes/dyn/libdartgpu_pack1.so
DARTGPU_LIB=$V timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/r05_ar_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r05_ar_tests.txt
tail -4 gpurun_out/r05_ar_tests.txt | cut -c1-300
grep -q "tests rc=0" gpurun_out/r05_ar_tests.txt || { kill $HB; exit 1; }
DARTGPU_LIB=$V timeout -k 10 300 python tests/probes/fuzz_parity.py 30 9900 > gpurun_out/r05_ar_fuzz_30_rounds.txt 2>&1 || { kill $HB; tail -5 gpurun_out/r05_ar_fuzz_30_rounds.txt; exit 1; }
tail -1 gpurun_out/r05_ar_fuzz_30_rounds.txt
for w in cfg5 human planted; do
  case $w in cfg5) A="--rlen 151 --spliced 0.3 --introns 20000"; VS="base pack1 base2 pack1b";; human) A="--genome-model human"; VS="base pack1";; planted) A=""; VS="base pack1";; esac
  for v in $VS; do
    case $v in pack1*) export DARTGPU_LIB=$V;; *) unset DARTGPU_LIB;; esac
    timeout -k 10 200 python bench.py $A --no-secondary --no-cpu-baseline --sustained-s 0 --human-like-budget 0 --steps 10 --warmup 2 > gpurun_out/r05_ar_${w}_$v.json 2> gpurun_out/r05_ar_${w}_$v.err || { kill $HB; exit 1; }
  done
done
kill $HB
python - <<'PY'
import json, glob
for w,vs in (("cfg5",("base","pack1","base2","pack1b")),("human",("base","pack1")),("planted",("base","pack1"))):
    for v in vs:
        d=json.loads(open("gpurun_out/r05_ar_%s_%s.json"%(w,v)).read().strip().splitlines()[-1])
        a=d["kernels_ms_one_batch_in_flight"]; 
        print(w, v, d["value"], "alone: k_report", a.get("k_report"), "k_reseed", a.get("k_reseed"), "k_pair", a.get("k_pair"))
PY
