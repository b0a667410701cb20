#!/bin/bash
# instruction counts of the seeding kernels per dispatch (PMC pass, one batch in flight)
export TMPDIR=/tmp
rm -rf gpurun_out/sr_pmc
rocprofv3 --pmc SQ_INSTS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS --output-format csv -d gpurun_out/sr_pmc -- python bench.py --no-cpu-baseline --inflight 1 --steps 1 --warmup 0 "$@" > /dev/null 2> gpurun_out/sr_pmc.err
python - <<'PY'
import csv, glob, collections
f = sorted(glob.glob('gpurun_out/sr_pmc/*/*counter_collection.csv'))[-1]
rows = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if 'k_seed' not in r['Kernel_Name']: continue
    rows.setdefault((int(r['Dispatch_Id']), r['Kernel_Name'][:24]), {})[r['Counter_Name']] = float(r['Counter_Value'])
for (d, k), c in list(rows.items())[-9:]:
    print("%-26s insts %7.1fM valu %6.1fM salu %6.1fM vmem %5.1fM lds %5.1fM waves %7d wave_cycles %8.1fM wait %8.1fM" % (k, c['SQ_INSTS']/1e6, c['SQ_INSTS_VALU']/1e6, c['SQ_INSTS_SALU']/1e6, c['SQ_INSTS_VMEM']/1e6, c['SQ_INSTS_LDS']/1e6, c['SQ_WAVES'], c['SQ_WAVE_CYCLES']/1e6, c['SQ_WAIT_ANY']/1e6))
PY
