"""profiles/traffic.json from the PMC passes of run_profile.sh: HBM-side bytes per launch per kernel.
bytes = (FETCH_SIZE + WRITE_SIZE) * 1024 (rocprofv3 reports KB), mean over the profiled dispatches; see _provenance."""
import csv, glob, json, os, sys
from collections import defaultdict
root, tag = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
def short(name):
    n = name.split("(")[0].replace("void ", "")
    return n.split("<")[0]
out = defaultdict(float)
for k, c in acc.items():
    if "k_" not in k or "FETCH_SIZE" not in c:
        continue
    b = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"]) + (sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"]) if "WRITE_SIZE" in c else 0.0)
    name = short(k)
    if name in ("k_seed_heavy", "k_seed_q", "k_seed_qf"):
        name = "k_seed"                      # bench.py times the two seeding kernels as one stage
    out[name] += b * 1024.0
doc = {"_provenance": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum (separate passes), profiles/run_profile.sh %s, "
                      "default bench workload, one batch in flight (2 M reads per launch), mean per launch; bytes = (FETCH_SIZE + WRITE_SIZE) * 1024; k_seed = k_seed_qf (or k_seed_q / k_seed) + "
                      "k_seed_heavy, k_report = mean of its two passes. Calibrated with known bytes (profiles/probes/fetch_calib.hip, "
                      "profiles/r04/e_fetch_size_calibration_random_accesses.txt): FETCH_SIZE is exact for random accesses of up to 64 bytes (one 64-byte fabric "
                      "request each) and reports half the bytes of 128-byte and wider accesses; the seeding, locate, pair and report kernels read 8 to 64 bytes per access, so the 2x "
                      "correction of MI355X_MICROARCH.md (wide coalesced streams) is NOT applied to them; the two streaming kernels (k_unpack, k_encode: 16 B per lane, coalesced) "
                      "are under-reported by their read half -- they are not the dominant kernel." % tag}
# the bench line of the profiled command says which kernel sources and which workload these passes belong to; bench.py reports
# `roofline.traffic` only when its own fingerprint equals this one
try:
    line = [l for l in open(os.path.join(root, "bench_trace.json")) if l.startswith("{")][-1]
    doc["_fingerprint"] = json.loads(line)["roofline"]["fingerprint"]
except Exception as e:
    doc["_fingerprint"] = None
    print("no fingerprint (%r): bench.py will not use this file" % (e,), file=sys.stderr)
doc.update({k: int(v) for k, v in sorted(out.items())})
json.dump(doc, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "traffic.json"), "w"), indent=1)
print(json.dumps(doc, indent=1))
