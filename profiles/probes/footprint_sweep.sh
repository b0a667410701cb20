#!/bin/bash
# Does the random 64-byte line rate depend on the footprint (address translation: the seeding stage looks things up in 5.4 + 118 GB)?
#   bash profiles/probes/footprint_sweep.sh > gpurun_out/footprint_sweep.txt
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
BIN=$ROOT/profiles/probes/fetch_calib_probe
[ -x $BIN ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 profiles/probes/fetch_calib.hip -o $BIN || exit 1
for gb in 3.1 16 49 118 200; do for w in 8 16; do for dep in 0 1; do $BIN $gb 64 $w 400 $dep || exit 1; done; done; done
for gb in 3.1 118; do
  D=/tmp/fp_$gb; rm -rf $D
  rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum --output-format csv -d $D -- $BIN $gb 64 8 400 0 > $D.out 2> $D.err || { echo "rocprofv3 failed"; tail -3 $D.err; continue; }
  grep requested $D.out
  python3 - "$D" <<'PY'
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True): rows += list(csv.DictReader(open(f)))
by = {}
for r in rows: by.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
if by: print("   counters of the measured dispatch:", {k: v for k, v in sorted(by[max(by)].items())})
PY
done
