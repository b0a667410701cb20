"""Stress probe of the single-pass scans (dart_amd/csrc/dg_scan.h): many small batches on many contexts, back to back.
    python tests/probes/scan_stress.py [batches per context = 10000] [contexts = 16] [reads per batch = 64000]
Round 2 saw one look-back in ~10^5 batches (12 contexts in flight) run out of its poll budget and papered over it with a re-run; since
round 3 the state words carry their run's epoch and are never zeroed.  Prints how many batches had to be run again because of a scan
(`reruns_scan_total`, must be 0), the rate, and compares every context's last records with the oracle."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
import common, oracle_py
from dart_amd import host, synth, index_build
per_ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
n_ctx = int(sys.argv[2]) if len(sys.argv) > 2 else 16
n_reads = int(sys.argv[3]) if len(sys.argv) > 3 else 64000
d = "/tmp/scan_stress"; os.makedirs(d, exist_ok=True)
g = synth.make_genome([3000000, 2000000], seed=91, repeat_scale=30.0, n_introns=200)
prefix = os.path.join(d, "idx")
index_build.build_index_from_genome(g, prefix)
ix = host.Index(prefix); orc = oracle_py.Oracle(prefix)
gpu = host.DartGPU(ix, host.default_params(paired=1, max_mismatch=5))
ctx = [gpu] + [gpu.clone() for _ in range(n_ctx - 1)]
sizes = (n_reads // 2, n_reads // 2 - 1234, n_reads // 2 - 7, n_reads // 4)
batches = []
for j, n in enumerate(sizes):
    m1, m2 = synth.make_reads(g, n, rlen=101, seed=92 + j, spliced_frac=0.1, indel_frac=0.04, n_frac=0.01)
    arr = host.interleave_pairs(m1, m2)
    batches.append((host.pack_reads(arr), host.pack_reads_2bit(arr)))
want = [orc.map_batch(orc.params(paired=1, max_mismatch=5), *b[0], threads=16) for b in batches]
out, errs, done = [None] * n_ctx, [], [0] * n_ctx
def work(k):
    try:
        for i in range(per_ctx):
            j = (i + k) % len(sizes)
            words, nlist = batches[j][1]
            res = ctx[k].map_batch_compact(words, nlist, 101)
            done[k] = i + 1
        out[k] = (j, res)
    except Exception as e:
        errs.append(e)
t0 = time.time()
th = [threading.Thread(target=work, args=(k,)) for k in range(n_ctx)]
for t in th: t.start()
while any(t.is_alive() for t in th):
    time.sleep(30)
    print("  %d batches so far, %.0f s" % (sum(done), time.time() - t0), flush=True)
for t in th: t.join()
dt = time.time() - t0
assert not errs, errs
reruns = sum(c.counters()["reruns_scan_total"] for c in ctx)
for k in range(n_ctx):
    j, res = out[k]
    common.assert_same(res, want[j])
print("%d batches of ~%d reads on %d contexts in %.1f s (%.0f batches/s): reruns_scan_total = %d; every context's last records equal the oracle's" %
      (per_ctx * n_ctx, n_reads, n_ctx, dt, per_ctx * n_ctx / dt, reruns))
sys.exit(1 if reruns else 0)
