# round 5, call ak: the persistent kernels' grid knobs on the spliced 2x151 shape (k_reseed's grid in per cent, k_report's waves per CU), same box
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( while true; do sleep 90; date +%s >> gpurun_out/r05_ak_heartbeat.txt; done ) &
HB=$!
W="--rlen 151 --spliced 0.3 --introns 20000 --no-secondary --no-cpu-baseline --sustained-s 0 --human-like-budget 0 --steps 10 --warmup 2"
for v in "base" "DG_RESEED_PCT=60" "DG_RESEED_PCT=80" "DG_RESEED_PCT=130" "DG_RESEED_PCT=160" "DG_REPORT_BPC=6" "DG_REPORT_BPC=10" "DG_RESEED_PCT=130 DG_REPORT_BPC=6" "base"; do
  if [ "$v" = base ]; then timeout -k 10 300 python bench.py $W > gpurun_out/r05_ak_one.json 2> gpurun_out/r05_ak_one.err || { kill $HB; exit 1; }
  else env $v timeout -k 10 300 python bench.py $W > gpurun_out/r05_ak_one.json 2> gpurun_out/r05_ak_one.err || { kill $HB; exit 1; }; fi
  python - "$v" <<'PY' >> gpurun_out/r05_ak_cfg5_grid_knobs.txt
import json, sys
d=json.loads(open("gpurun_out/r05_ak_one.json").read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], "k_reseed alone/in flight", d["kernels_ms_one_batch_in_flight"].get("k_reseed"), d["kernels_ms"].get("k_reseed"), "k_report", d["kernels_ms_one_batch_in_flight"].get("k_report"), d["kernels_ms"].get("k_report"))
PY
  tail -1 gpurun_out/r05_ak_cfg5_grid_knobs.txt
done
kill $HB
