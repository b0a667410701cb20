# round 5, call d: the whole GPU suite (launch chain 21 -> 15 per batch: no k_batch_begin, k_unpack_listed in k_prep, k_order_jobs in k_order), then the three bench shapes
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r05_d_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r05_d_tests.txt
tail -5 gpurun_out/r05_d_tests.txt
grep -q "tests rc=0" gpurun_out/r05_d_tests.txt || exit 1
timeout -k 10 300 python bench.py --genome-model human --no-secondary --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/r05_d_human.json 2> gpurun_out/r05_d_human.err; echo "human rc=$?"
timeout -k 10 300 python bench.py --no-secondary --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/r05_d_planted.json 2> gpurun_out/r05_d_planted.err; echo "planted rc=$?"
python - <<'PY'
import json
for f in ("human","planted"):
    try:
        d=json.loads(open("gpurun_out/r05_d_%s.json"%f).read().strip().splitlines()[-1])
        print(f, d["value"], d.get("value_repeats")); print(" inflight", {k:round(v,2) for k,v in d["kernels_ms"].items()}); print(" alone", {k:round(v,2) for k,v in d["kernels_ms_one_batch_in_flight"].items()})
    except Exception as e: print(f, "failed", e)
PY
