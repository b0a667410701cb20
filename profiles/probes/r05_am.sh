# round 5, call am: d_identify_sj with its two genome windows in registers (variant library) beside the shipped build: parity module (the spliced golden cases are in it),
# then the spliced 2x151 shape, the human-like genome and the planted genome on the same box
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( while true; do sleep 90; date +%s >> gpurun_out/r05_am_heartbeat.txt; done ) &
HB=$!
V=$GRAFT_REPO_ROOT/profiles/probes/dyn/libdartgpu_sj.so
DARTGPU_LIB=$V timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r05_am_parity.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r05_am_parity.txt
tail -2 gpurun_out/r05_am_parity.txt
grep -q "tests rc=0" gpurun_out/r05_am_parity.txt || { kill $HB; exit 1; }
for w in cfg5 human planted; do
  case $w in cfg5) A="--rlen 151 --spliced 0.3 --introns 20000";; human) A="--genome-model human";; planted) A="";; esac
  for v in shipped sj shipped2 sj2; do
    case $v in shipped*) unset DARTGPU_LIB;; *) export DARTGPU_LIB=$V;; esac
    timeout -k 10 300 python bench.py $A --no-secondary --no-cpu-baseline --sustained-s 0 --human-like-budget 0 --steps 10 --warmup 2 > gpurun_out/r05_am_${w}_$v.json 2> gpurun_out/r05_am_${w}_$v.err || { kill $HB; exit 1; }
  done
done
kill $HB
python - <<'PY'
import json
for w in ("cfg5","human","planted"):
    for v in ("shipped","sj","shipped2","sj2"):
        d=json.loads(open("gpurun_out/r05_am_%s_%s.json"%(w,v)).read().strip().splitlines()[-1])
        print(w, v, d["value"], "k_report alone/in flight", d["kernels_ms_one_batch_in_flight"].get("k_report"), d["kernels_ms"].get("k_report"))
PY
