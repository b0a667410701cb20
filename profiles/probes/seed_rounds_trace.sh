#!/bin/bash
# per-dispatch durations of the seeding kernels (one batch in flight), from a rocprofv3 kernel trace
export TMPDIR=/tmp
rm -rf gpurun_out/sr_trace
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/sr_trace -- python bench.py --no-cpu-baseline --inflight 1 --steps 2 --warmup 1 "$@" > gpurun_out/sr.json 2> gpurun_out/sr.err
python - <<'PY'
import csv, glob
f = sorted(glob.glob('gpurun_out/sr_trace/*/*kernel_trace.csv'))[-1]
rows = [r for r in csv.DictReader(open(f)) if 'k_seed' in r['Kernel_Name'] or 'k_encode' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
last = rows[-9:]
for r in last:
    print("%-28s %8.3f ms  grid %s" % (r['Kernel_Name'][:28], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6, r.get('Grid_Size', r.get('Grid_Size_X', '?'))))
PY
