"""debug probe: one large batch against the same reads mapped in pieces (records must not depend on the batching); prints which reads differ.
usage: python tests/probes/whole_vs_parts.py [chr20|grch38] [pairs]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench
from dart_amd import host, synth
which = sys.argv[1] if len(sys.argv) > 1 else "chr20"
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
label, names, lens = bench.genome_spec(which)
prefix, g = bench.prepare_index("/tmp/dart_bench_cache", (names, lens), 0, lambda: None)
ix = host.Index(prefix)
gpu = host.DartGPU(ix, host.default_params(paired=1, max_mismatch=5))
m1, m2 = synth.make_reads(g, pairs, rlen=101, seed=1002, sub_rate=0.01, indel_frac=0.02, n_frac=0.002)
arr = host.interleave_pairs(m1, m2)
small = gpu.map_batch(*host.pack_reads(arr[:2 * 200000]))          # (sizes the context for a smaller batch first, as the test module does)
for attempt in range(3):
    whole = gpu.map_batch(*host.pack_reads(arr))
    cuts = [0, 2 * (pairs * 31 // 100 + 1), 2 * (pairs * 7 // 9), 2 * pairs]
    parts = [gpu.map_batch(*host.pack_reads(arr[a:b])) for a, b in zip(cuts[:-1], cuts[1:])]
    bad = set()
    for f in ("score", "sub_score", "mis_num", "mapq", "n_rep", "best", "n_sj"):
        d = np.nonzero(np.concatenate([p.reads[f] for p in parts]) != whole.reads[f])[0]
        bad |= set(d.tolist())
    bad = sorted(bad)
    print("attempt", attempt, "reads that differ:", len(bad), bad[:20], "counters", {k: v for k, v in gpu.counters().items() if k in ("batch_runs", "general_path_units", "wave_chained_units", "seeds")})
    for r in bad[:3]:
        print("   read", r, "whole", whole.reads[r], "\n        part ", np.concatenate([p.reads for p in parts])[r])
