# round 5, call ap: the whole GPU suite, smoke, 60 fuzz rounds and a 2000-step strict soak on the sources with the splice windows in registers and d_nw_group<16>
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( while true; do sleep 90; date +%s >> gpurun_out/r05_ap_heartbeat.txt; done ) &
HB=$!
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r05_ap_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r05_ap_tests.txt
tail -4 gpurun_out/r05_ap_tests.txt | cut -c1-300
grep -q "tests rc=0" gpurun_out/r05_ap_tests.txt || { kill $HB; exit 1; }
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_ap_smoke.txt 2>&1 || { kill $HB; tail -5 gpurun_out/r05_ap_smoke.txt; exit 1; }
tail -1 gpurun_out/r05_ap_smoke.txt | cut -c1-120
timeout -k 10 900 python tests/probes/fuzz_parity.py 60 9300 > gpurun_out/r05_ap_fuzz_60_rounds.txt 2>&1 || { kill $HB; tail -5 gpurun_out/r05_ap_fuzz_60_rounds.txt; exit 1; }
tail -1 gpurun_out/r05_ap_fuzz_60_rounds.txt
DART_BENCH_STRICT=1 timeout -k 10 300 python bench.py --no-secondary --no-cpu-baseline --sustained-s 0 --human-like-budget 0 --steps 2000 --warmup 3 > gpurun_out/r05_ap_soak_2000_steps.json 2> gpurun_out/r05_ap_soak_2000_steps.err; echo "soak rc=$?"
kill $HB
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05_ap_soak_2000_steps.json").read().strip().splitlines()[-1])
print("soak: value", d["value"], "steps", d["steps"], {k:v for k,v in d["counters_per_launch"].items() if "rerun" in k})
PY
