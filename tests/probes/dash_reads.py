"""Probe (GPU box): the five reads with a literal '-' that led to the fix of DESIGN 6, round 5 item 18 (a one-lane d_nw in front of a candidate's large pair overwrote its traceback bits), alone, single-end, with their dashes as they are / as N / as a base / as a lower-case base: library against oracle, CIGAR and score.
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import common, oracle_py
from dart_amd import host, synth
wd = "/tmp/dash_dbg"; os.makedirs(wd, exist_ok=True)
oracle_py.build()
c = common.build_case("pe101_spliced", wd)
ix = host.Index(c["prefix"]); gpu = host.DartGPU(ix); orc = oracle_py.Oracle(c["prefix"])
R = [
 b"CGGTTTTGTTTCGATAAACGTAGGTAAATTCTACGTCGGCCGTTC-GCTCCCATACCATGAACTTTTCAATTTTTGC-AAAATGGCAGTGGCCGCTTGTGG",
 b"ACCGCTAGC-CATCAGAAAAACCTCGCAAAGATCACTTGCCTCGTGATGAAGGGTCCTCTGCGGACCCTGACTATGGGGCGGTAGGGCAAGTCGAGGTGGTCCAAAGGGAAGGGTGAAGAAGAAAGATGTTCGGGGACTTG-CTAGTACGGTCTCGTCCGGAATCTGAGGCGGGGCTGCTCGCGGCATAACCCTTCAATGATGTTGTCGGACTAATTACTTTTGTAAGGTCGATTAATCGGTTCTGGGTT",
 b"TTGCCGGCGGCACCGGCACATGCGGTATCACGCTCGTG-AGGCCGACGTCTCGTCACGGTTGCTGGTGGGTCTTGACCTTTTCTTGC-AACCTCGGGGCATAAAAATTAGTCAGAATTTACTCGCATACTGAAACAGAAGCCTGCCCGCATCAGAGTGAGGGGTGTAACACGGTGTCTAGTTGAAACCTAGAAAGAGGGGGTGTGGGGACCTATACCGAGGTATTGCACTCGGCCCCCAAAACTTAGAGA",
 b"CAGTTAGAGCCGCTCAGTGCGTCTAACGGCCTACCCGTGAGTCGACATGTCCATGCA-GCCAAGAGGAGTAGTGCCGCACTTTGGTTACGAACTGGCCG-ACAATGGCACAATATGGAGGTGCTTTTTAGCCACTCGGTAGGAACGAGCACCACTCATATGTCGGGCCAAATACGTTAATCCTAGTAGCACGACAGGTACACAGTGCCTTTCCCCCCCAAAAAATCCCGAGGTTACTATTGAAATTAGAG",
 b"GGCCTTGTCCCAGTAAATACAAACGTTAAACCCAATAAGCGACGCTGCTTCAATTCAAAGAGAATCTTGCTCTGTAAACAATCCGGATGAACATGACGGC-GGACGTAATCTCGTTA-ACGCCGTTATCATCTGCTCGCGATAGGTGGCGTAGTGACGTCAGACGGCATGCGGCATCGACTATGTATAACTGCCACTACCCGGCTCTTGCTTCACCCCAAGACCCTTTTTCTGA-TATCCATGCCCTTTA",
]
OPS = "MIDNS"
def cig(rep, cg, k):
    o, n = int(rep["cigar_off"][k]), int(rep["n_cigar"][k])
    return "".join("%d%s" % (int(x) >> 4, OPS[int(x) & 15]) for x in cg[o:o + n])
variants = []
for r in R:
    variants += [r, r.replace(b"-", b"N"), r.replace(b"-", b"A"), r.replace(b"-", b"c")]
so, rl, flat = host.pack_reads(variants)
for mis in (12, 30):
    gpu.set_params(host.default_params(paired=0, max_mismatch=mis))
    res = gpu.map_batch(so, rl, flat)
    reads, rep, cg, sj = orc.map_batch(orc.params(paired=0, max_mismatch=mis), so, rl, flat)
    for k in range(len(variants)):
        o1, o2 = int(reads["rep_off"][k]), int(res.reads["rep_off"][k])
        print("mis", mis, "read", k // 4, ["as is", "N", "A", "c"][k % 4], "oracle", int(reads["score"][k]), cig(rep, cg, o1), "| gpu", int(res.reads["score"][k]), cig(res.reports, res.cigar, o2), "" if (int(reads["score"][k]) == int(res.reads["score"][k]) and cig(rep, cg, o1) == cig(res.reports, res.cigar, o2)) else "  <-- DIFFERS")
    sd = orc.seeds(orc.params(paired=0, max_mismatch=mis), variants[4]) if mis == 12 else None
    if sd is not None: print("oracle seeds of read 1 as is:", sd)
    if mis == 12:
        ps = gpu.probe_seeds(*host.pack_reads([variants[4]]))
        print("gpu probe_seeds of read 1 as is:", [np.asarray(x).tolist() for x in ps])
