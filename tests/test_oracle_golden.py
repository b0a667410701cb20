"""CPU suite: the oracle (oracle/dart_oracle.c) against the committed golden vectors that were
produced by the reference's own object code (tests/golden/make_golden.py), and our index builder
against the digests of the reference indexer's files."""
import gzip, os
import numpy as np
import pytest
import common, oracle_py
from dart_amd import host

CASES = sorted(common.MANIFEST["cases"])


@pytest.fixture(scope="module")
def oracles(workdir):
    out = {}
    for name in CASES:
        c = common.build_case(name, workdir)
        out[name] = (c, oracle_py.Oracle(c["prefix"]), host.Index(c["prefix"]))
    return out


@pytest.mark.parametrize("name", CASES)
def test_index_builder_matches_reference_indexer(name, workdir):
    c = common.build_case(name, workdir)
    for ext, want in c["index_sha"].items():
        assert common.sha(c["prefix"] + "." + ext) == want, "index file .%s differs from the reference bwt_index output" % ext


@pytest.mark.parametrize("name", CASES)
def test_oracle_sam_and_junctions_match_reference(name, oracles):
    c, orc, ix = oracles[name]
    so, rl, flat = host.pack_reads(c["reads"])
    for run in c["runs"]:
        p, h = common.parse_flags(run["flags"])
        reads, rep, cig, sj = orc.map_batch(orc.params(paired=int(c["spec"]["paired"]), **p), so, rl, flat, threads=4)
        text, junc = common.records_to_text(c, p, h, reads, rep, cig, sj, ix)
        want = common.golden_sam(run["base"])
        assert text == want, common.first_diff(text, want)
        assert junc == common.golden_junctions(run["base"])


@pytest.mark.parametrize("name", CASES)
def test_oracle_cli_prints_the_reference_statistics(name, workdir):
    """the statistics block at the end of a run's stdout (Mapping.cpp:812-822: mapped / paired / unique / multiple / unmapped / junctions):
    the oracle's command line against what the reference's own counters printed for the same run (tests/golden/*.stats.txt)"""
    import subprocess
    from dart_amd import synth
    oracle_py.build()
    c = common.build_case(name, workdir)
    d = os.path.join(workdir, "stats_" + name); os.makedirs(d, exist_ok=True)
    synth.write_fastq(os.path.join(d, "1.fq"), c["m1"], 1)
    files = ["-f", "1.fq"]
    if c["spec"]["paired"]:
        synth.write_fastq(os.path.join(d, "2.fq"), c["m2"], 2); files += ["-f2", "2.fq"]
    for run in c["runs"]:
        r = subprocess.run([oracle_py.ORACLE_CLI, "-i", c["prefix"]] + files + ["-o", "o.sam", "-j", "o.j", "-t", "2"] + run["flags"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True)
        assert common.stats_block(r.stdout) == common.golden_stats(run["base"]), (run["base"], r.stdout[-600:])
        assert open(os.path.join(d, "o.sam")).read() == common.golden_sam(run["base"])


def test_oracle_threads_do_not_change_results(oracles):
    c, orc, ix = oracles["pe101_spliced"]
    so, rl, flat = host.pack_reads(c["reads"])
    p = orc.params(paired=1, max_mismatch=5)
    a = orc.map_batch(p, so, rl, flat, threads=1)
    b = orc.map_batch(p, so, rl, flat, threads=7)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_oracle_nw_known_answers():
    orc_lib = oracle_py.Oracle.__new__(oracle_py.Oracle)
    oracle_py.build()
    import ctypes as C
    orc_lib.lib = C.CDLL(oracle_py.LIB)
    orc_lib.lib.orc_nw.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
    n = 0
    for line in gzip.open(os.path.join(common.GOLDEN, "nw_known_answers.tsv.gz"), "rt"):
        a, b, o1, o2 = line.rstrip("\n").split("\t")
        assert orc_lib.nw(a.encode(), b.encode()) == (o1.encode(), o2.encode()), (a, b)
        n += 1
    assert n > 1500


def test_oracle_seeds_match_reference_stage_dump(oracles):
    """S1/S2 lines of the stage dump = IdentifySeedPairs output of the reference per read."""
    for name in CASES:
        c, orc, ix = oracles[name]
        base = c["runs"][0]["base"]
        p, _ = common.parse_flags(c["runs"][0]["flags"])
        k = {"S1": 0, "S2": 1}
        idx = {"S1": 0, "S2": 0}
        checked = 0
        for line in gzip.open(os.path.join(common.GOLDEN, base + ".stages.gz"), "rt"):
            f = line.split()
            if f[0] not in k:
                continue
            i = idx[f[0]]; idx[f[0]] += 1
            if i % 5:
                continue
            r = (2 * i + k[f[0]]) if c["spec"]["paired"] else i
            rp, sl, gp = orc.seeds(orc.params(**p), c["reads"][r].tobytes())
            want = [tuple(int(x) for x in t.split(":")) for t in f[3:]]
            assert list(zip(rp.tolist(), sl.tolist(), gp.tolist())) == want
            checked += 1
        assert checked > 100


def test_oracle_bwt_search_known_answers(oracles):
    import ctypes as C
    c, orc, ix = oracles["pe101_spliced"]
    orc.lib.orc_bwt_search.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    p = orc.params()
    nt4 = np.full(256, 4, np.uint8)
    for i, ch in enumerate(b"ACGT"):
        nt4[ch] = i; nt4[ch + 32] = i
    lines = gzip.open(os.path.join(common.GOLDEN, "bwt_search_known_answers.txt.gz"), "rt").read().strip().split("\n")
    assert len(lines) == 60
    for r, line in enumerate(lines):
        enc = np.ascontiguousarray(nt4[c["m1"][r]])
        for tok in line.split():
            head, *locs = tok.split(",")
            start, ln, freq = (int(x) for x in head.split(":"))
            out = np.zeros(128, np.uint64); l = C.c_int(0)
            got = orc.lib.orc_bwt_search(orc.ix, C.byref(p), enc.ctypes.data, start, len(enc), C.byref(l), out.ctypes.data)
            assert got == freq
            if freq:
                assert l.value == ln and out[:freq].tolist() == [int(x) for x in locs]


def test_bucketed_suffix_sorter_gives_the_same_index(workdir, monkeypatch):
    """the memory-lean sorter used for >= 1.6 G-symbol texts (dart_amd/index_build.py::suffix_array_bucketed) against the
    plain prefix doubling, on a repeat-rich two-chromosome genome (several refinement rounds)"""
    from dart_amd import synth, index_build
    g = synth.make_genome([300000, 200000], seed=3, repeat_scale=30.0)
    a, b = os.path.join(workdir, "sa_plain"), os.path.join(workdir, "sa_bucketed")
    index_build.build_index_from_genome(g, a, device="cpu")
    monkeypatch.setenv("DART_SA_BUCKETED", "1")
    index_build.build_index_from_genome(g, b, device="cpu")
    for ext in ("bwt", "sa", "pac", "ann", "amb"):
        assert open(a + "." + ext, "rb").read() == open(b + "." + ext, "rb").read(), ext


def test_oracle_cli_matches_reference_on_reads_with_odd_characters(workdir):
    """Single-end reads with a literal '-' (a gap to AddNewCigarElements, tools.cpp:49-104), lower case, N and IUPAC letters, -mis 12: the oracle's command line against the SAM
    and junctions the reference's object code wrote for the same reads (tests/golden/odd_characters.*, made by tests/golden/make_odd_characters.py).  Half of the reads
    hold a dash; such reads are what the library's string forms exist for."""
    import json, hashlib, subprocess
    oracle_py.build()
    c = common.build_case("pe101_spliced", workdir)
    meta = json.load(open(os.path.join(common.GOLDEN, "odd_characters.json")))
    seqs = common.odd_character_reads(c["genome"])
    assert len(seqs) == meta["reads"] and hashlib.sha256(b"\n".join(seqs)).hexdigest() == meta["reads_sha256"], "the read generator drifted from the golden inputs"
    d = os.path.join(workdir, "odd_characters"); os.makedirs(d, exist_ok=True)
    common.write_se_fastq(os.path.join(d, "odd.fq"), seqs)
    subprocess.run([oracle_py.ORACLE_CLI, "-i", c["prefix"], "-f", "odd.fq", "-mis", "12", "-o", "orc.sam", "-j", "orc.j", "-t", "4"], cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
    want = gzip.open(os.path.join(common.GOLDEN, "odd_characters.mis12.sam.gz"), "rt").read()
    got = open(os.path.join(d, "orc.sam")).read()
    assert got == want, common.first_diff(got, want)
    assert open(os.path.join(d, "orc.j")).read() == open(os.path.join(common.GOLDEN, "odd_characters.mis12.junctions.tab")).read()
