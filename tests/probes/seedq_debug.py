"""debug: the medium parity batch, every flag set, queue kernel and lane-per-read kernel against the oracle"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import common, oracle_py
from dart_amd import synth, index_build, host
if os.environ.get("DARTGPU_LIB"): host.LIB_PATH = os.environ["DARTGPU_LIB"]; print("library", host.LIB_PATH)
g = synth.make_genome([2000000, 1000000], seed=31, repeat_scale=50.0, n_introns=400)
os.makedirs("/tmp/sqd", exist_ok=True)
prefix = "/tmp/sqd/medium"
index_build.build_index_from_genome(g, prefix)
oracle_py.build()
ix = host.Index(prefix); gpu = host.DartGPU(ix); orc = oracle_py.Oracle(prefix)
m1, m2 = synth.make_reads(g, 20000, rlen=101, seed=32, spliced_frac=0.2, indel_frac=0.05, n_frac=0.01)
so, rl, flat = host.pack_reads(host.interleave_pairs(m1, m2))
for flags in (["-mis", "5"],):
    p, _ = common.parse_flags(flags)
    ores = orc.map_batch(orc.params(paired=1, **p), so, rl, flat, threads=16)
    for leg in ("0",):
        os.environ["DG_SEED_LEGACY"] = leg
        gpu.set_params(host.default_params(paired=1, **p))
        res = gpu.map_batch(so, rl, flat)
        bad = set()
        for f in ores[0].dtype.names:
            if f in ("sj_off", "rep_off"): continue
            bad |= set(np.nonzero(ores[0][f] != res.reads[f])[0].tolist())
        print(flags, "legacy" if leg == "1" else "queue", "reads that differ:", sorted(bad)[:10], gpu.counters()["reseed_calls"])
        for r in sorted(bad)[:3]:
            print("   read", r, "oracle", ores[0][r], "gpu", res.reads[r])
            print("   ", bytes(flat[so[r]:so[r] + rl[r]]).decode())
            ro = ores[0][r]; rg = res.reads[r]
            print("    oracle reports", ores[1][ro["rep_off"]:ro["rep_off"] + ro["n_rep"]])
            print("    gpu    reports", res.reports[rg["rep_off"]:rg["rep_off"] + rg["n_rep"]])
