# round 5, call au: the round's FINAL kernel sources (commit cde0354).  Kernel trace + PMC passes for the three workloads (the traffic files must carry these sources'
# fingerprint), the driver's default bench command, 60 fuzz rounds, a 2000-step strict soak.  (The GPU suite and smoke on these sources: call at.)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( while true; do sleep 60; date +%s >> gpurun_out/r05_au_heartbeat.txt; done ) &
HB=$!
T0=$(date +%s)
PROFILE_PASSES="fetch tcc inst" timeout -k 10 300 bash profiles/run_profile.sh r05au_planted > gpurun_out/r05_au_prof_planted.log 2>&1; echo "planted rc=$? at $(( $(date +%s) - T0 )) s"
PROFILE_PASSES="fetch tcc inst" timeout -k 10 300 bash profiles/run_profile.sh r05au_human --genome-model human > gpurun_out/r05_au_prof_human.log 2>&1; echo "human rc=$? at $(( $(date +%s) - T0 )) s"
PROFILE_PASSES="fetch tcc inst" timeout -k 10 300 bash profiles/run_profile.sh r05au_spliced --rlen 151 --spliced 0.3 --introns 20000 > gpurun_out/r05_au_prof_spliced.log 2>&1; echo "spliced rc=$? at $(( $(date +%s) - T0 )) s"
# the traffic files of these passes, so that the default command below reports roofline.traffic
python - <<'PY'
import json
for w,f in (("planted","traffic.json"),("human","traffic_human.json"),("spliced","traffic_spliced.json")):
    t=open("gpurun_out/prof_r05au_%s/traffic.txt"%w).read()
    json.dump(json.loads(t[t.index("{"):]), open("profiles/"+f,"w"), indent=1)
PY
( time python bench.py ) > gpurun_out/r05_au_bench_default.json 2> gpurun_out/r05_au_bench_default.err; echo "bench rc=$? at $(( $(date +%s) - T0 )) s"
grep "^real" gpurun_out/r05_au_bench_default.err
timeout -k 10 300 python tests/probes/fuzz_parity.py 60 11300 > gpurun_out/r05_au_fuzz_60_rounds.txt 2>&1; echo "fuzz rc=$? at $(( $(date +%s) - T0 )) s"; tail -1 gpurun_out/r05_au_fuzz_60_rounds.txt
DART_BENCH_STRICT=1 timeout -k 10 200 python bench.py --no-secondary --no-cpu-baseline --sustained-s 0 --human-like-budget 0 --steps 2000 --warmup 3 > gpurun_out/r05_au_soak_2000_steps.json 2> gpurun_out/r05_au_soak_2000_steps.err; echo "soak rc=$? at $(( $(date +%s) - T0 )) s"
kill $HB
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05_au_bench_default.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value","value_sustained","value_human_like","value_cli_end_to_end_grch38","value_ascii_input","value_full_records","value_device_resident")})
print("roofline", {k: d["roofline"].get(k) for k in ("kernel","achieved","frac","traffic")})
s=json.loads(open("gpurun_out/r05_au_soak_2000_steps.json").read().strip().splitlines()[-1])
print("soak: value", s["value"], "steps", s["steps"])
PY
