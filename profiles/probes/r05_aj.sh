# round 5, call aj: k_reseed's bitmap filter on 15 / 16 bits of the 8-mer instead of 14 (fewer false candidates keep the divergent probe loop running): parity module, then the
# spliced 2x151 shape and the human-like genome, each variant beside the shipped build on the same box
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( while true; do sleep 90; date +%s >> gpurun_out/r05_aj_heartbeat.txt; done ) &
HB=$!
for v in flt15 flt16; do
  DARTGPU_LIB=$GRAFT_REPO_ROOT/profiles/probes/dyn/libdartgpu_$v.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r05_aj_parity_$v.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r05_aj_parity_$v.txt
  tail -2 gpurun_out/r05_aj_parity_$v.txt
  grep -q "tests rc=0" gpurun_out/r05_aj_parity_$v.txt || { kill $HB; exit 1; }
done
for v in shipped flt15 flt16 shipped2; do
  case $v in shipped*) unset DARTGPU_LIB;; *) export DARTGPU_LIB=$GRAFT_REPO_ROOT/profiles/probes/dyn/libdartgpu_$v.so;; esac
  timeout -k 10 300 python bench.py --rlen 151 --spliced 0.3 --introns 20000 --no-secondary --no-cpu-baseline --sustained-s 0 --human-like-budget 0 --steps 10 --warmup 2 > gpurun_out/r05_aj_cfg5_$v.json 2> gpurun_out/r05_aj_cfg5_$v.err || { kill $HB; exit 1; }
done
kill $HB
python - <<'PY'
import json
for v in ("shipped","flt15","flt16","shipped2"):
    d=json.loads(open("gpurun_out/r05_aj_cfg5_%s.json"%v).read().strip().splitlines()[-1])
    c=d["counters_per_launch"]
    print(v, d["value"], "k_reseed alone/in flight", d["kernels_ms_one_batch_in_flight"].get("k_reseed"), d["kernels_ms"].get("k_reseed"), "trips", c.get("k_reseed_trips"))
PY
