"""CPU suite: the accuracy scorers (dart_amd/evaluate.py) -- hand-made known answers, the golden SAM /
junction files of the spliced case against the generator's truth, and (where oracle/_ref exists) the reference's own
eva / SJ_Eva programs run on the same files."""
import os, re, subprocess
import numpy as np
import pytest
import common, oracle_py
from dart_amd import evaluate

REF_EVA = os.path.join(oracle_py.ORACLE_DIR, "_ref", "eva")
REF_SJ = os.path.join(oracle_py.ORACLE_DIR, "_ref", "SJ_Eva")
REF_FLUX = os.path.join(oracle_py.ORACLE_DIR, "_ref", "FluxEva")


def test_eva_known_answers():
    chroms = {"c1": "ACGTACGTACGTACGTACGT", "c2": "TTTTTTTTTT"}
    sam = ["@SQ\tSN:c1\tLN:20",
           "r1\t0\tc1\t1\t50\t8M\t*\t0\t0\tACGTACGT\tIIIIIIII",                 # 8/8
           "r2\t0\tc1\t1\t50\t4M1I4M\t*\t0\t0\tACGTGACGT\tIIIIIIIII",           # 8 of 9 columns
           "r3\t0\tc1\t1\t50\t4M2D4M\t*\t0\t0\tACGTGTAC\tIIIIIIII",             # 8 of 10 columns
           "r4\t4\t*\t0\t0\t*\t*\t0\t0\tACGT\tIIII",                            # unaligned
           "r5\t0\tc1\t3\t50\t2S4M4N2M\t*\t0\t0\tNNGTACGT\tIIIIIIII",           # S and N: GTAC vs GTAC, then GT vs c1[10:12]=GT -> 6/6
           "r5\t256\tc1\t3\t50\t8M\t*\t0\t0\tNNGTACGT\tIIIIIIII", "r5\t256\tc1\t3\t50\t8M\t*\t0\t0\tNNGTACGT\tIIIIIIII"]   # third hit of r5 is ignored
    r = evaluate.eva(sam, chroms)
    assert r["records"] == 6 and r["aligned"] == 5
    want = [1000, 1000 * 8 // 9, 1000 * 8 // 10, 1000, 0]      # (soft clips are not alignment columns; r5's second record matches nowhere)
    assert abs(r["avg_identity"] - sum(want) / 5 / 1000.0) < 1e-9
    j = evaluate.sj_eva([("c1", 100, 200), ("c1", 104, 204), ("c1", 105, 200), ("c9", 1, 2)], [("c1", 100, 200), ("c1", 500, 900)])
    assert j["annotated"] == 2 and j["predicted"] == 4 and j["true_junctions"] == 2 and abs(j["recall"] - 0.5) < 1e-9


def _spliced_case(workdir):
    c = common.build_case("pe101_spliced", workdir)
    g = c["genome"]
    asc = g.ascii()
    chroms = {n: asc[o:o + l].tobytes().decode() for n, o, l in zip(g.names, g.offsets, g.lengths)}
    base = c["runs"][1]["base"]                         # the -mis 5 run
    return c, g, chroms, common.golden_sam(base), common.golden_junctions(base)


def test_scorers_on_golden_case(workdir):
    c, g, chroms, sam, junc = _spliced_case(workdir)
    r = evaluate.eva(sam.split("\n"), chroms)
    assert r["records"] >= 2 * c["spec"]["npairs"] and r["sensitivity"] > 0.9 and r["avg_identity"] > 0.98
    truth = []
    ends = np.cumsum(g.lengths)
    for s, ilen, _ in g.introns:
        ci = int(np.searchsorted(ends, s, side="right"))
        truth.append((g.names[ci], int(s - g.offsets[ci]) + 1, int(s - g.offsets[ci] + ilen)))
    pred = [tuple(l.split()[:3]) for l in junc.split("\n") if l.strip()]
    j = evaluate.sj_eva(pred, truth)
    assert j["predicted"] > 50 and j["precision"] > 0.9, j


@pytest.mark.skipif(not (os.path.exists(REF_EVA) and os.path.exists(REF_SJ)), reason="oracle/_ref scorers not built (no /root/reference here)")
def test_scorers_match_reference_programs(workdir):
    c, g, chroms, sam, junc = _spliced_case(workdir)
    d = os.path.join(workdir, "eva_ref"); os.makedirs(d, exist_ok=True)
    g.write_fasta(os.path.join(d, "hg38.fa"))             # the reference program opens this name in its working directory
    open(os.path.join(d, "out.sam"), "w").write(sam)
    out = subprocess.run([REF_EVA, "out.sam"], cwd=d, capture_output=True, text=True).stderr
    m = re.findall(r"sensitivity = (\d+) / (\d+) = ([0-9.]+), AvgSeqIdy = ([0-9.]+)", out)[-1]
    r = evaluate.eva(sam.split("\n"), chroms)
    assert (r["aligned"], r["records"]) == (int(m[0]), int(m[1]))
    assert "%.3f" % (r["avg_identity"] + 0.0005) == m[3]    # the reference prints value + 0.0005 with three decimals
    truth = []
    ends = np.cumsum(g.lengths)
    for s, ilen, _ in g.introns:
        ci = int(np.searchsorted(ends, s, side="right"))
        truth.append((g.names[ci], int(s - g.offsets[ci]) + 1, int(s - g.offsets[ci] + ilen)))
    open(os.path.join(d, "junctions.txt"), "w").write("".join("%s\t%d\t%d\n" % t for t in truth))
    open(os.path.join(d, "pred.tab"), "w").write(junc)
    out = subprocess.run([REF_SJ, "pred.tab"], cwd=d, capture_output=True, text=True).stdout
    acc = int(re.search(r"Acc = (\d+)", out).group(1)); n_rep = int(re.search(r"# of Reported SJ = (\d+)", out).group(1))
    pred = [tuple(l.split()[:3]) for l in junc.split("\n") if l.strip()]
    j = evaluate.sj_eva(pred, truth)
    assert (j["annotated"], j["predicted"]) == (acc, n_rep)


def _flux_sam(workdir, n_keep=4000, seed=5):
    """the golden SAM of the spliced case with Flux-simulator read names ("<chr>:<left>-<right>W:..."): the region is the record's own POS
    shifted by a random amount, so that some records fall inside it and some outside; plus the cases the program treats specially --
    records without CIGAR, MAPQ 0, more than two records per name, a name on another chromosome, names without 'W' or '-'"""
    c, g, chroms, sam, junc = _spliced_case(workdir)
    rng = np.random.default_rng(seed)
    out = []
    body = [l for l in sam.split("\n") if l and l[0] != "@"][:n_keep]
    out += [l for l in sam.split("\n") if l.startswith("@")]
    for i in range(0, len(body) - 1, 2):
        f1, f2 = body[i].split("\t"), body[i + 1].split("\t")
        pos = int(f1[3])
        left = max(0, pos - int(rng.integers(0, 300)) + (400 if rng.random() < 0.15 else 0))
        right = left + int(rng.integers(100, 600))
        chr_ = f1[2] if (f1[2] != "*" and rng.random() < 0.9) else g.names[int(rng.integers(0, len(g.names)))]
        style = i // 2 % 11
        name = "%s:%d-%dW:%d:%d" % (chr_, left, right, i, 7)
        if style == 9: name = "%s:%d-%d" % (chr_, left, right)                 # no 'W'
        if style == 10: name = "%s:%d" % (chr_, left)                           # no '-'
        for f in (f1, f2):
            f[0] = name
            if style == 7: f[4] = "0"
            out.append("\t".join(f))
        if style == 8:                                                          # a third and a fourth record of the same name: not counted
            out.append("\t".join(f1)); out.append("\t".join(f2))
    return "\n".join(out) + "\n"


def test_flux_eva_known_answers():
    sam = ["@SQ\tSN:c1\tLN:1000",
           "c1:100-200W:x\t0\tc1\t150\t50\t8M\t*\t0\t0\tACGTACGT\tIIIIIIII",      # right
           "c1:100-200W:x\t0\tc1\t201\t50\t8M\t*\t0\t0\tACGTACGT\tIIIIIIII",      # outside
           "c1:100-200W:x\t0\tc1\t150\t50\t8M\t*\t0\t0\tACGTACGT\tIIIIIIII",      # third record of the name: not counted
           "c1:300-400W:y\t4\t*\t0\t0\t*\t*\t0\t0\tACGT\tIIII",                   # empty
           "c1:300-400W:y\t0\tc1\t300\t0\t4M\t*\t0\t0\tACGT\tIIII",               # MAPQ 0: left out
           "c2:5-9W:z\t0\tc1\t7\t50\t4M\t*\t0\t0\tACGT\tIIII",                    # another chromosome
           "c1:5-9W:z\t0\tc1\t9\t50\t4M\t*\t0\t0\tACGT\tIIII"]                    # right (inclusive bound)
    r = evaluate.flux_eva(sam)
    assert (r["right"], r["scored"], r["records"], r["empty"], r["mapq0"]) == (2, 4, 6, 1, 1) and r["accuracy_percent"] == 50.0


@pytest.mark.skipif(not os.path.exists(REF_FLUX), reason="oracle/_ref/FluxEva not built (no /root/reference here)")
def test_flux_eva_matches_reference_program(workdir):
    d = os.path.join(workdir, "flux_ref"); os.makedirs(d, exist_ok=True)
    for seed in (5, 6):
        sam = _flux_sam(workdir, seed=seed)
        open(os.path.join(d, "flux.sam"), "w").write(sam)
        out = subprocess.run([REF_FLUX, "flux.sam"], cwd=d, capture_output=True, text=True).stdout
        m = re.search(r"Acc = (\d+) / (\d+) = ([0-9.]+)\s*$", out.split("\r")[-1])
        assert m, out
        r = evaluate.flux_eva(sam.split("\n"))
        assert (r["right"], r["scored"]) == (int(m.group(1)), int(m.group(2))), (r, out[-200:])
        assert "%.2f" % r["accuracy_percent"] == m.group(3)
        assert r["right"] > 200 and r["scored"] - r["right"] > 100 and r["mapq0"] > 100 and r["empty"] > 0
