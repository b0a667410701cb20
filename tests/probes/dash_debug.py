"""Probe for reads with a literal '-' (tests/test_gpu_parity.py::test_gpu_literal_dashes_in_read_gaps_and_segment_pairs): maps the test's reads single-end and paired through the
library and the oracle and prints the records of the reads that differ."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import common, oracle_py
from dart_amd import host, synth
wd = sys.argv[1] if len(sys.argv) > 1 else "/tmp/dash_dbg"
os.makedirs(wd, exist_ok=True)
oracle_py.build()
c = common.build_case("pe101_spliced", wd)
ix = host.Index(c["prefix"]); gpu = host.DartGPU(ix); orc = oracle_py.Oracle(c["prefix"])
rng = np.random.default_rng(77)
seqs = []
for rlen, n_pairs, seed in ((101, 6000, 501), (250, 2500, 502)):
    m1, m2 = synth.make_reads(c["genome"], n_pairs, rlen=rlen, seed=seed, spliced_frac=0.6, indel_frac=0.3, n_frac=0.0)
    for i in range(n_pairs):
        for m in (m1, m2):
            s = bytearray(m[i].tobytes())
            k = i % 8
            if k < 3:
                for q in rng.integers(0, rlen, size=k + 1): s[int(q)] = ord("-")
            elif k == 3:
                a = int(rng.integers(20, rlen - 70)); w = int(rng.integers(26, 60))
                s[a:a + w] = bytes(rng.choice(list(b"ACGT"), w).astype(np.uint8)); s[a + w // 2] = ord("-")
            elif k == 4:
                a = int(rng.integers(20, rlen - 70)); w = int(rng.integers(26, 60))
                s[a:a + w] = bytes(rng.choice(list(b"ACGT"), w).astype(np.uint8))
            elif k == 5:
                for q in rng.integers(0, rlen, size=4): s[int(q)] = s[int(q)] | 0x20
                s[int(rng.integers(0, rlen))] = ord("N")
            seqs.append(bytes(s))
so, rl, flat = host.pack_reads(seqs)
OPS = "MIDNS"
def cig(rep, cg, k):
    o, n = int(rep["cigar_off"][k]), int(rep["n_cigar"][k])
    return "".join("%d%s" % (int(x) >> 4, OPS[int(x) & 15]) for x in cg[o:o + n])
for paired, mis in ((1, 12), (0, 12)):
    gpu.set_params(host.default_params(paired=paired, max_mismatch=mis))
    res = gpu.map_batch(so, rl, flat)
    reads, rep, cg, sj = orc.map_batch(orc.params(paired=paired, max_mismatch=mis), so, rl, flat)
    bad = [r for r in range(len(reads)) if any(reads[f][r] != res.reads[f][r] for f in reads.dtype.names if f != "sj_off")]
    print("paired", paired, "reads that differ:", len(bad), bad[:12])
    print(" read fields", reads.dtype.names, "report fields", rep.dtype.names)
    for r in bad[:6]:
        print(" read", r, seqs[r].decode())
        print("  oracle", {f: int(reads[f][r]) for f in reads.dtype.names})
        print("  gpu   ", {f: int(res.reads[f][r]) for f in reads.dtype.names})
        for name, R, P, C in (("oracle", reads, rep, cg), ("gpu", res.reads, res.reports, res.cigar)):
            names = R.dtype.names
            ofs = [f for f in names if "rep" in f and "off" in f] or [f for f in names if f.endswith("_off") and f != "sj_off"]
            cnt = [f for f in names if f in ("n_rep", "CanNum", "n_reports", "can_num")]
            if not ofs or not cnt: continue
            o, n = int(R[ofs[0]][r]), int(R[cnt[0]][r])
            for k in range(o, o + n):
                print("   %s rep %d: %s cigar %s" % (name, k - o, {f: int(P[f][k]) for f in P.dtype.names if f not in ("cigar_off",)}, cig(P, C, k)))
