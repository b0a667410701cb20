# round 5, call u: the command-line leg with the WHOLE job mapped by the CPU command line too (10 M pairs, ~100 s on 16 cores): SAM and junctions compared by digest
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( while true; do sleep 90; date +%s >> gpurun_out/r05_u_heartbeat.txt; done ) &
HB=$!
timeout -k 10 1000 python bench.py --steps 5 --warmup 2 --no-secondary --sustained-s 0 --human-like-budget 0 --cli-full-parity --cli-gz-pairs 0 > gpurun_out/r05_u_bench.json 2> gpurun_out/r05_u_bench.err; echo "bench rc=$?"
kill $HB
grep "^\[bench\]" gpurun_out/r05_u_bench.err | cut -c1-600 | tail -8
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05_u_bench.json").read().strip().splitlines()[-1])
print(json.dumps(d["cli_end_to_end_grch38"].get("whole_job_parity"), indent=1))
PY
# ... then 60 fuzz rounds on the final code (fresh genomes, read sets, flag sets; every record against the oracle, through the ASCII, packed and compact entry points),
# and a 3000-step run that treats any batch run again by a look-back as fatal
timeout -k 10 900 python tests/probes/fuzz_parity.py 60 9100 > gpurun_out/r05_u_fuzz_60_rounds.txt 2>&1; echo "fuzz rc=$?"; tail -2 gpurun_out/r05_u_fuzz_60_rounds.txt | cut -c1-200
DART_BENCH_STRICT=1 timeout -k 10 300 python bench.py --no-secondary --no-cpu-baseline --steps 3000 --warmup 3 > gpurun_out/r05_u_soak_3000_steps.json 2> gpurun_out/r05_u_soak_3000_steps.err; echo "soak rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05_u_soak_3000_steps.json").read().strip().splitlines()[-1])
print("soak: value", d["value"], "steps", d["steps"], "reruns", {k:v for k,v in d["counters_per_launch"].items() if "rerun" in k})
PY
