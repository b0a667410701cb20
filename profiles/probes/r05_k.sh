# round 5, call k: the profile of the spliced 2x151 shape (BASELINE configs[4]) on the round's final kernel sources, and its bench line with the free-context item distribution
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( while true; do sleep 90; date +%s >> gpurun_out/r05_k_heartbeat.txt; done ) &
HB=$!
PROFILE_PASSES="fetch tcc inst" bash profiles/run_profile.sh r05_spliced --rlen 151 --spliced 0.3 --introns 20000 > gpurun_out/r05_k_prof_spliced.log 2>&1; echo "prof rc=$?"
kill $HB
timeout -k 10 400 python bench.py --rlen 151 --spliced 0.3 --introns 20000 --no-secondary --no-cpu-baseline --sustained-s 10 --steps 10 --warmup 2 > gpurun_out/r05_k_cfg5.json 2> gpurun_out/r05_k_cfg5.err; echo "cfg5 rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05_k_cfg5.json").read().strip().splitlines()[-1])
print("cfg5", d["value"], d.get("value_sustained"), d.get("sustained")); print(" inflight", {k:round(v,2) for k,v in d["kernels_ms"].items()}); print(" alone", {k:round(v,2) for k,v in d["kernels_ms_one_batch_in_flight"].items()})
PY
