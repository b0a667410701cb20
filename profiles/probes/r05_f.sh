# round 5, call f: (1) do page walks show up in the memory-side read requests of random 16-byte look-ups over the aids' footprints; (2) the rate over time of a long run;
# (3) batch composition x contexts on the planted genome (the step is 10 M pairs; how it is cut into batches is ours)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
BIN=profiles/probes/fetch_calib_probe
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 profiles/probes/fetch_calib.hip -o $BIN || exit 1
{
for gb in 3.1 69 118; do for b in 16 64; do
  D=/tmp/pw_${gb}_$b; rm -rf $D
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $D -- $BIN $gb $b 8 400 0 > $D.out 2> $D.err || { echo "rocprofv3 failed"; tail -3 $D.err; continue; }
  grep requested $D.out
  python3 - "$D" <<'PY'
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True): rows += list(csv.DictReader(open(f)))
by = {}
for r in rows: by.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
if by: print("   counters of the measured dispatch (52.4 M accesses):", {k: v for k, v in sorted(by[max(by)].items())})
PY
done; done
} > gpurun_out/r05_f_pagewalk_rdreq.txt 2>&1
cat gpurun_out/r05_f_pagewalk_rdreq.txt
bash profiles/probes/sustained.sh r05f_planted 3000 > gpurun_out/r05_f_sustained_planted.txt 2>&1; tail -40 gpurun_out/r05_f_sustained_planted.txt
V=""
for pb in 500000:20 1000000:10 2000000:5 2500000:4; do for i in 8 12 16; do V="$V p${pb%%:*}_i$i:-:DART_BENCH_PAIRS=${pb%%:*},DART_BENCH_BATCHES=${pb##*:},DART_BENCH_INFLIGHT=$i"; done; done
bash profiles/probes/variants.sh r05f_composition_planted "$V"
