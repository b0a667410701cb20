# round 5, call h: the driver's own sequence -- the whole GPU suite, smoke, the default bench command (with its CPU legs, the reference-object-code leg, the sustained pass, the human-like leg)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r05_h_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r05_h_tests.txt
tail -5 gpurun_out/r05_h_tests.txt
grep -q "tests rc=0" gpurun_out/r05_h_tests.txt || exit 1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_h_smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r05_h_smoke.txt
( time python bench.py --steps 20 --warmup 5 ) > gpurun_out/r05_h_bench_default.json 2> gpurun_out/r05_h_bench_default.err; echo "bench rc=$?"
grep "^\[bench\]" gpurun_out/r05_h_bench_default.err | cut -c1-300 | tail -30
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05_h_bench_default.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value","value_repeats","value_sustained","value_human_like","value_cli_end_to_end_grch38","value_ascii_input","value_full_records","value_device_resident")})
print("sustained", d.get("sustained")); print("phases", d.get("phases_s"))
print("cpu", {k:v for k,v in d["cpu_baseline"].items() if k!="sample"})
print("human", {k: d.get("human_like",{}).get(k) for k in ("value","sustained","skipped")})
print("roofline", {k: d["roofline"].get(k) for k in ("kernel","achieved","frac","traffic","achieved_basis")})
PY
