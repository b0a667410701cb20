# round 5, call l: k_seed_heavy's replay jumps from hit to hit; every thread of k_order takes re-seeding jobs -- the parity file, the GRCh38-sized tests, then human-like and spliced bench lines
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_grch38_human.py tests/test_gpu_grch38.py -m gpu -x -q > gpurun_out/r05_l_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r05_l_tests.txt
tail -4 gpurun_out/r05_l_tests.txt
grep -q "tests rc=0" gpurun_out/r05_l_tests.txt || exit 1
timeout -k 10 300 python bench.py --genome-model human --no-secondary --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/r05_l_human.json 2> gpurun_out/r05_l_human.err; echo "human rc=$?"
timeout -k 10 400 python bench.py --rlen 151 --spliced 0.3 --introns 20000 --no-secondary --no-cpu-baseline --steps 10 --warmup 2 > gpurun_out/r05_l_cfg5.json 2> gpurun_out/r05_l_cfg5.err; echo "cfg5 rc=$?"
python - <<'PY'
import json
for f in ("human","cfg5"):
    try:
        d=json.loads(open("gpurun_out/r05_l_%s.json"%f).read().strip().splitlines()[-1])
        print(f, d["value"]); print(" inflight", {k:round(v,2) for k,v in d["kernels_ms"].items()}); print(" alone", {k:round(v,2) for k,v in d["kernels_ms_one_batch_in_flight"].items()})
    except Exception as e: print(f, "failed", e)
PY
