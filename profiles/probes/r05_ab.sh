# round 5, call ab: the default bench command with the HIP index builder (the driver's bench step; the suite and smoke ran in call aa)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time python bench.py ) > gpurun_out/r05_ab_bench_default.json 2> gpurun_out/r05_ab_bench_default.err; echo "bench rc=$?"
grep "^\[bench\]\|^real" gpurun_out/r05_ab_bench_default.err | cut -c1-300 | tail -30
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05_ab_bench_default.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value","value_repeats","value_sustained","value_human_like","value_cli_end_to_end_grch38","value_ascii_input","value_full_records","value_device_resident")})
print("phases", d.get("phases_s"))
print("human", {k: d.get("human_like",{}).get(k) for k in ("value","sustained","skipped","phases_s")})
print("roofline", {k: d["roofline"].get(k) for k in ("kernel","achieved","frac","traffic","achieved_basis")})
PY
