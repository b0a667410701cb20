# round 5, call e: (1) which counters does rocprofv3 offer here; (2) the human-like genome in flight with smaller look-up aids (is the memory side -- translations over
# 118 GB of aids, random lines -- what a dozen batches in flight run into?); (3) PMC passes of the seeding stage on the human-like genome, one batch in flight
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocprofv3 -L > gpurun_out/r05_e_counters_avail.txt 2>&1 || rocprofv3 --list-avail > gpurun_out/r05_e_counters_avail.txt 2>&1
grep -o -i -E "\b(TCC_EA0?_[A-Z0-9_]*|TCP_UTCL1_[A-Z_]*|TCC_(REQ|HIT|MISS|READ|WRITE|ATOMIC|PROBE)[A-Z0-9_]*|TCP_TCC_[A-Z_]*|TCP_TA_[A-Z_]*|UTCL2[A-Z_0-9]*|TCP_GATE_EN[0-9]*[A-Z_]*)\b" gpurun_out/r05_e_counters_avail.txt | sort -u > gpurun_out/r05_e_counters_memory_side.txt
V="base:-: k15:-:DG_KTAB_K=15 k14:-:DG_KTAB_K=14 sa2:-:DG_SA_DENSE=2 sa4:-:DG_SA_DENSE=4 k14sa4:-:DG_KTAB_K=14,DG_SA_DENSE=4 if14:-:DART_BENCH_INFLIGHT=14 base2:-:"
bash profiles/probes/variants.sh r05e_human_aids "$V" --genome-model human
PMC="--steps 1 --warmup 0 --batches 2 --inflight 1 --no-cpu-baseline --no-secondary --genome-model human"
pass() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/r05e_pmc/$name -- python3 bench.py $PMC > /dev/null 2> gpurun_out/r05e_pmc_$name.err; echo "pass $name rc=$?"; }
pass ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
pass tcc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_WRITE_sum
pass utcl TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_PERMISSION_MISS_sum
pass tcp TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
python3 - <<'PY'
import csv, glob, os, collections
for d in sorted(glob.glob("gpurun_out/r05e_pmc/*")):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0][:24]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[(k, row["Counter_Name"])] += 1
    print("==", os.path.basename(d))
    for k in sorted(acc):
        if any(s in k for s in ("k_seed", "k_pair", "k_locate", "k_report", "k_chain", "k_prep")):
            print("  %-26s" % k, {c: round(v / max(1, n[(k, c)]), 1) for c, v in acc[k].items()})
PY
rm -rf gpurun_out/r05e_pmc
