set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "reseed_windows or spliced_2x151 or golden or edge or scan_timeout or counters or medium_batch" > gpurun_out/r05_a_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r05_a_tests.txt
tail -5 gpurun_out/r05_a_tests.txt
timeout -k 10 400 python bench.py --genome-model human --no-secondary --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/r05_a_human.json 2> gpurun_out/r05_a_human.err; echo "human rc=$?"
timeout -k 10 400 python bench.py --rlen 151 --spliced 0.3 --introns 20000 --no-secondary --no-cpu-baseline --steps 10 --warmup 2 > gpurun_out/r05_a_cfg5.json 2> gpurun_out/r05_a_cfg5.err; echo "cfg5 rc=$?"
python - <<'PY'
import json
for f in ("human","cfg5"):
    try:
        d=json.loads(open("gpurun_out/r05_a_%s.json"%f).read().strip().splitlines()[-1])
        print(f, d["value"], d.get("value_repeats")); print(" inflight", {k:round(v,2) for k,v in d["kernels_ms"].items()}); print(" alone", {k:round(v,2) for k,v in d["kernels_ms_one_batch_in_flight"].items()})
        print(" ctr", {k:v for k,v in d.get("counters_per_launch",{}).items() if "reseed" in k})
    except Exception as e: print(f, "failed", e)
PY
