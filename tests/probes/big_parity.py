"""GPU vs oracle on a big synthetic genome: where do the records differ?  python profiles/probes/big_parity.py GENOME_BP"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench, oracle_py
from dart_amd import host, synth
glen = int(float(sys.argv[1]))
prefix, g = bench.prepare_index("/tmp/dart_bench_cache", glen, 0, lambda: None)
m1, m2 = synth.make_reads(g, 20000, rlen=101, seed=1000, sub_rate=0.01, indel_frac=0.02, n_frac=0.002)
so, rl, flat = host.pack_reads(host.interleave_pairs(m1, m2))
ix = host.Index(prefix)
print("seq_len", ix.seq_len, "primary", ix.primary, "L2", ix.L2)
orc = oracle_py.Oracle(prefix)
o_reads, o_rep, o_cig, o_sj = orc.map_batch(orc.params(paired=1, max_mismatch=5), so, rl, flat, threads=16)
print("oracle counters", orc.counters)
for env in ({},):
    os.environ.pop("DG_KTAB_K", None); os.environ.pop("DG_SA_DENSE", None)
    os.environ.update(env)
    gpu = host.DartGPU(ix, host.default_params(paired=1, max_mismatch=5))
    res = gpu.map_batch(so, rl, flat)
    c = gpu.counters()
    bad = np.nonzero((o_reads["score"] != res.reads["score"]) | (o_reads["n_rep"] != res.reads["n_rep"]) | (o_reads["best"] != res.reads["best"]))[0]
    print(env, "reads differing in score/n_rep/best:", len(bad), "of", len(o_reads), "first", bad[:8],
          "| gpu steps", c["steps"], "orc", orc.counters["n_2occ4"], "| gpu sa", c["sa_lookups"], "orc", orc.counters["n_sa"], "| lf", c["lf_steps"], orc.counters["n_lf"])
    nrep = min(len(o_rep), len(res.reports))
    pb = np.nonzero(o_rep["pos"][:nrep] != res.reports["pos"][:nrep])[0]
    print("   reports", len(o_rep), len(res.reports), "pos differs at", len(pb), pb[:5], [(int(o_rep["pos"][i]), int(res.reports["pos"][i])) for i in pb[:5]])
    for r in bad[:3]:
        print("   read", r, "oracle", o_reads[r], "gpu", res.reads[r])
    gpu.close()
