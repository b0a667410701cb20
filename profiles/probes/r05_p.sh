# round 5, call p (again): k_chain_heavy sorts up to 64 seeds per mate in registers -- parity file + the human-like GRCh38-sized test, bench lines, instruction counts
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_grch38_human.py -m gpu -x -q > gpurun_out/r05_p_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r05_p_tests.txt
tail -4 gpurun_out/r05_p_tests.txt
grep -q "tests rc=0" gpurun_out/r05_p_tests.txt || exit 1
timeout -k 10 300 python bench.py --genome-model human --no-secondary --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/r05_p_human.json 2> gpurun_out/r05_p_human.err; echo "human rc=$?"
timeout -k 10 300 python bench.py --no-secondary --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/r05_p_planted.json 2> gpurun_out/r05_p_planted.err; echo "planted rc=$?"
python - <<'PY'
import json
for f in ("human","planted"):
    try:
        d=json.loads(open("gpurun_out/r05_p_%s.json"%f).read().strip().splitlines()[-1])
        print(f, d["value"]); print(" inflight", {k:round(v,2) for k,v in d["kernels_ms"].items()}); print(" alone", {k:round(v,2) for k,v in d["kernels_ms_one_batch_in_flight"].items()})
    except Exception as e: print(f, "failed", e)
PY
bash profiles/pmc_inst.sh r05p_human --genome-model human > gpurun_out/r05_p_pmc_inst_human.txt 2>&1; tail -12 gpurun_out/r05_p_pmc_inst_human.txt | cut -c1-250
