# round 5, call ad: the k_pair dynamic-LDS variant under the whole parity module, then its rates beside the shipped build's (planted, human-like; k_pair alone / in flight)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
DYN=$GRAFT_REPO_ROOT/profiles/probes/dyn/libdartgpu_kpair_dynamic_lds.so
DARTGPU_LIB=$DYN timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r05_ad_parity_dynamic.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r05_ad_parity_dynamic.txt
tail -4 gpurun_out/r05_ad_parity_dynamic.txt | cut -c1-300
grep -q "tests rc=0" gpurun_out/r05_ad_parity_dynamic.txt || exit 1
for v in static dynamic; do
  if [ $v = dynamic ]; then export DARTGPU_LIB=$DYN; else unset DARTGPU_LIB; fi
  timeout -k 10 300 python bench.py --no-secondary --no-cpu-baseline --sustained-s 0 --human-like-budget 0 --steps 20 --warmup 3 > gpurun_out/r05_ad_planted_$v.json 2> gpurun_out/r05_ad_planted_$v.err || exit 1
  timeout -k 10 300 python bench.py --genome-model human --no-secondary --no-cpu-baseline --sustained-s 0 --steps 20 --warmup 3 > gpurun_out/r05_ad_human_$v.json 2> gpurun_out/r05_ad_human_$v.err || exit 1
done
python - <<'PY'
import json
for w in ("planted","human"):
    for v in ("static","dynamic"):
        d=json.loads(open("gpurun_out/r05_ad_%s_%s.json"%(w,v)).read().strip().splitlines()[-1])
        k=d.get("kernels_ms") or {}
        print(w, v, d["value"], d.get("value_repeats"), "k_pair alone/in flight:", (d.get("kernels_ms_one_batch_in_flight") or {}).get("k_pair"), (d.get("kernels_ms") or {}).get("k_pair"))
PY
