# round 5, call at: the product build with the traceback consumers on the bits (TbWalk: d_process_pair_tb, d_gap_right_tb / d_gap_left_tb) beside the build before it
# (profiles/probes/dyn/libdartgpu_pack1.so = commit 0c7e0e6): the whole GPU suite, smoke and 40 fuzz rounds on the product build, then spliced 2x151 / human-like / planted A / B
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( while true; do sleep 60; date +%s >> gpurun_out/r05_at_heartbeat.txt; done ) &
HB=$!
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/r05_at_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r05_at_tests.txt
tail -4 gpurun_out/r05_at_tests.txt | cut -c1-300
grep -q "tests rc=0" gpurun_out/r05_at_tests.txt || { kill $HB; exit 1; }
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_at_smoke.txt 2>&1 || { kill $HB; tail -5 gpurun_out/r05_at_smoke.txt; exit 1; }
tail -1 gpurun_out/r05_at_smoke.txt | cut -c1-120
timeout -k 10 300 python tests/probes/fuzz_parity.py 40 10300 > gpurun_out/r05_at_fuzz_40_rounds.txt 2>&1 || { kill $HB; tail -5 gpurun_out/r05_at_fuzz_40_rounds.txt; exit 1; }
tail -1 gpurun_out/r05_at_fuzz_40_rounds.txt
V=$GRAFT_REPO_ROOT/profiles/probes/dyn/libdartgpu_pack1.so
for w in cfg5 human planted; do
  case $w in cfg5) A="--rlen 151 --spliced 0.3 --introns 20000"; VS="before tbw before2 tbw2";; human) A="--genome-model human"; VS="before tbw";; planted) A=""; VS="before tbw";; esac
  for v in $VS; do
    case $v in before*) export DARTGPU_LIB=$V;; *) unset DARTGPU_LIB;; esac
    timeout -k 10 200 python bench.py $A --no-secondary --no-cpu-baseline --sustained-s 0 --human-like-budget 0 --steps 10 --warmup 2 > gpurun_out/r05_at_${w}_$v.json 2> gpurun_out/r05_at_${w}_$v.err || { kill $HB; exit 1; }
  done
done
kill $HB
python - <<'PY'
import json
for w,vs in (("cfg5",("before","tbw","before2","tbw2")),("human",("before","tbw")),("planted",("before","tbw"))):
    for v in vs:
        d=json.loads(open("gpurun_out/r05_at_%s_%s.json"%(w,v)).read().strip().splitlines()[-1])
        a=d["kernels_ms_one_batch_in_flight"]
        print(w, v, d["value"], "alone: k_report", a.get("k_report"), "k_reseed", a.get("k_reseed"), "k_pair", a.get("k_pair"))
PY
