#!/bin/bash
# Usage (on the GPU box): bash profiles/probes/fetch_calib.sh > gpurun_out/fetch_calib.txt
# Timing first (no profiler), then one `rocprofv3 --pmc` pass per configuration and counter group; prints requested bytes beside FETCH_SIZE.
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
BIN=$ROOT/profiles/probes/fetch_calib_probe
[ -x $BIN ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 profiles/probes/fetch_calib.hip -o $BIN || exit 1
GB=${CALIB_GB:-3.1}
echo "== timing, footprint $GB GB (the Occ array of a GRCh38-sized text) =="
for dep in 0 1; do for b in 16 64 128 256; do for w in 4 8 16; do $BIN $GB $b $w 400 $dep || exit 1; done; done; done
echo "== counters: the second dispatch of each run is the measured one (8 waves/CU, 400 iterations, independent) =="
for b in 16 64 128 256; do
  for grp in "FETCH_SIZE" "TCC_MISS_sum TCC_HIT_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
    D=/tmp/calib_${b}_$(echo $grp | cut -c1-6)
    rm -rf $D
    rocprofv3 --pmc $grp --output-format csv -d $D -- $BIN $GB $b 8 400 0 > $D.out 2> $D.err || { echo "rocprofv3 failed for $grp"; tail -3 $D.err; continue; }
    grep requested $D.out
    python3 - "$D" "$b" <<'PY'
import csv, glob, os, sys
root, b = sys.argv[1], int(sys.argv[2])
rows = []
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
by = {}
for r in rows:
    by.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
if by:
    d = by[max(by)]
    print("   counters of the measured dispatch:", {k: v for k, v in sorted(d.items())})
    if "FETCH_SIZE" in d:
        print("   FETCH_SIZE * 1024 = %.0f bytes" % (d["FETCH_SIZE"] * 1024))
PY
  done
done
