# round 5, call an: d_nw_group<16> (65 .. 128-column alignments four at a time) in the product build beside the build without it (profiles/probes/dyn/libdartgpu_sj_only.so):
# the parity module (nw known answers: 150+ vectors of that width; spliced goldens), then the spliced 2x151 shape and the human-like genome on the same box
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( while true; do sleep 90; date +%s >> gpurun_out/r05_an_heartbeat.txt; done ) &
HB=$!
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r05_an_parity.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r05_an_parity.txt
tail -2 gpurun_out/r05_an_parity.txt
grep -q "tests rc=0" gpurun_out/r05_an_parity.txt || { kill $HB; exit 1; }
V=$GRAFT_REPO_ROOT/profiles/probes/dyn/libdartgpu_sj_only.so
for w in cfg5 human; do
  case $w in cfg5) A="--rlen 151 --spliced 0.3 --introns 20000";; human) A="--genome-model human";; esac
  for v in before g16 before2 g16b; do
    case $v in before*) export DARTGPU_LIB=$V;; *) unset DARTGPU_LIB;; esac
    timeout -k 10 300 python bench.py $A --no-secondary --no-cpu-baseline --sustained-s 0 --human-like-budget 0 --steps 10 --warmup 2 > gpurun_out/r05_an_${w}_$v.json 2> gpurun_out/r05_an_${w}_$v.err || { kill $HB; exit 1; }
  done
done
kill $HB
python - <<'PY'
import json
for w in ("cfg5","human"):
    for v in ("before","g16","before2","g16b"):
        d=json.loads(open("gpurun_out/r05_an_%s_%s.json"%(w,v)).read().strip().splitlines()[-1])
        print(w, v, d["value"], "k_report alone/in flight", d["kernels_ms_one_batch_in_flight"].get("k_report"), d["kernels_ms"].get("k_report"))
PY
