"""Shared helpers of the test-suite: golden cases rebuilt from their seeds, flag parsing, SAM text."""
from __future__ import annotations

import gzip, hashlib, json, os, sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLDEN = os.path.join(HERE, "golden")
sys.path.insert(0, ROOT)
from dart_amd import synth, index_build, host, sam  # noqa: E402

MANIFEST = json.load(open(os.path.join(GOLDEN, "manifest.json")))
_cache = {}


def sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def build_case(name: str, workdir: str):
    """Regenerates a golden case's inputs from its seeds; builds its index with our own builder."""
    if name in _cache:
        return _cache[name]
    spec = MANIFEST["cases"][name]
    g = synth.make_genome(spec["lengths"], seed=spec["gseed"], repeat_scale=spec["rscale"], n_introns=spec["nintr"])
    m1, m2 = synth.make_reads(g, spec["npairs"], rlen=spec["rlen"], seed=spec["rseed"], spliced_frac=spec["spliced"],
                              paired=spec["paired"], sub_rate=spec["sub"], indel_frac=0.05, n_frac=0.01)
    h = hashlib.sha256()
    for a in (g.codes, m1, m2):
        if a is not None:
            h.update(np.ascontiguousarray(a).tobytes())
    assert h.hexdigest() == MANIFEST["manifest"][name]["inputs_sha256"], "synthetic generator drifted from the golden inputs"
    prefix = os.path.join(workdir, name)
    index_build.build_index_from_genome(g, prefix, device="cpu")
    arr = host.interleave_pairs(m1, m2) if spec["paired"] else m1
    n = arr.shape[0]
    if spec["paired"]:
        headers = ["r%d" % (i // 2) for i in range(n)]
    else:
        headers = ["r%d" % i for i in range(n)]
    case = dict(name=name, spec=spec, genome=g, m1=m1, m2=m2, prefix=prefix, reads=arr, headers=headers,
                seqs=[arr[i].tobytes().decode() for i in range(n)], quals=["I" * arr.shape[1]] * n,
                runs=MANIFEST["manifest"][name]["runs"], index_sha=MANIFEST["manifest"][name]["index_sha256"])
    _cache[name] = case
    return case


def parse_flags(flags):
    """DART command-line flags -> (path params, host-only options)  (main.cpp:169-192)."""
    p = dict(max_gaps=5, max_dup=100, max_intron=500000, min_intron=5, max_mismatch=0, multi_hit=0, all_sj=0)
    h = dict(unique=False)
    i = 0
    while i < len(flags):
        f = flags[i]
        if f == "-mis": p["max_mismatch"] = int(flags[i + 1]); i += 1
        elif f == "-max_dup": p["max_dup"] = min(max(int(flags[i + 1]), 100), 10000); i += 1
        elif f == "-max_intron": p["max_intron"] = max(int(flags[i + 1]), 100000); i += 1
        elif f == "-min_intron": p["min_intron"] = int(flags[i + 1]); i += 1
        elif f == "-m": p["multi_hit"] = 1
        elif f == "-all_sj": p["all_sj"] = 1
        elif f == "-unique": h["unique"] = True
        else: raise ValueError(f)
        i += 1
    return p, h


def golden_sam(base):
    return gzip.open(os.path.join(GOLDEN, base + ".sam.gz"), "rb").read().decode()


def golden_stats(base):
    return open(os.path.join(GOLDEN, base + ".stats.txt")).read()


def stats_block(stdout_bytes):
    """the statistics block a run prints at its end (Mapping.cpp:812-822): the '\t# of ...' lines of its stdout, without the junction file's name -- what tests/golden/*.stats.txt hold of the reference's stdout"""
    import re
    lines = [l for l in stdout_bytes.decode("latin1").replace("\r", "\n").split("\n") if l.startswith("\t# of")]
    return "".join(re.sub(r" \(file: .*\)$", "", l) + "\n" for l in lines)


def golden_junctions(base):
    return open(os.path.join(GOLDEN, base + ".junctions.tab")).read()


def records_to_text(case, params, hostopts, reads, reports, cigar, sj, ix: host.Index):
    body = sam.format_records(case["headers"], case["seqs"], case["quals"], reads, reports, cigar, ix.names,
                              paired=case["spec"]["paired"], multi_hit=bool(params["multi_hit"]), unique_only=hostopts["unique"])
    text = sam.sam_header(ix.names, ix.chr_len) + body
    junc = sam.junction_table(sj, ix.names, ix.chr_off, ix.chr_len, ix.l_pac)
    return text, junc


def first_diff(a: str, b: str):
    la, lb = a.split("\n"), b.split("\n")
    for i, (x, y) in enumerate(zip(la, lb)):
        if x != y:
            fx, fy = x.split("\t"), y.split("\t")
            cols = [k for k in range(min(len(fx), len(fy))) if fx[k] != fy[k]]
            return "line %d differs in columns %s:\n  got      %s | %s\n  expected %s | %s" % (
                i, cols, " ".join(fx[:9]), " ".join(fx[11:]), " ".join(fy[:9]), " ".join(fy[11:]))
    return "length differs: %d vs %d lines" % (len(la), len(lb))


# ---- record comparison (GPU result object vs the oracle's arrays), shared by the GPU tests, bench-style probes and __graft_entry__.smoke ----
def cigars_of(reports, cigar):
    """the CIGAR ops of every report, in report order (cigar_off/n_cigar dereferenced: the ABI fixes what a report's ops are,
    not where in the op array they lie)"""
    off = reports["cigar_off"].astype(np.int64); k = reports["n_cigar"].astype(np.int64)
    idx = np.repeat(off - np.concatenate([[0], np.cumsum(k)[:-1]]), k) + np.arange(int(k.sum()))
    return cigar[idx] if len(idx) else np.zeros(0, np.uint32)


def assert_same(res, ores):
    reads, rep, cig, sj = ores
    for f in reads.dtype.names:
        if f != "sj_off":
            assert np.array_equal(reads[f], res.reads[f]), "read field %s differs at %s" % (f, np.nonzero(reads[f] != res.reads[f])[0][:5])
    has_sj = reads["n_sj"] > 0                            # (sj_off says where a read's tuples are; it means nothing for a read without any)
    assert np.array_equal(reads["sj_off"][has_sj], res.reads["sj_off"][has_sj])
    assert len(rep) == len(res.reports)
    for f in rep.dtype.names:
        if f != "cigar_off":
            assert np.array_equal(rep[f], res.reports[f]), "report field %s differs at %s" % (f, np.nonzero(rep[f] != res.reports[f])[0][:5])
    assert len(cig) == len(res.cigar)
    a, b = cigars_of(rep, cig), cigars_of(res.reports, res.cigar)
    assert np.array_equal(a, b), "CIGAR ops differ, first at flattened op %s" % (np.nonzero(a != b)[0][:5] if len(a) == len(b) else "(length)")
    assert np.array_equal(sj, res.sj)


def write_holes_fasta(path: str) -> None:
    """The FASTA of tests/golden/index_holes.json (made by tests/golden/make_index_holes.py with the reference's bwt_index): three records with
    header comments, lower-case bases, runs of N / n and other ambiguity codes (adjacent runs of different characters, a run at the start and at
    the end of a record, a run that ends one record while the next starts with the same character), lines of different widths, an empty line,
    '\\r\\n' line ends in one record -- what bntseq.c:104-156 turns into holes and random bases."""
    rng = np.random.default_rng(4242)
    def bases(k):
        return "".join("ACGT"[i] for i in rng.integers(0, 4, k))
    rec1 = "NNNNN" + bases(700) + "NNNNNNNNNNRRRRNNN" + bases(1200).lower() + "n" + bases(333) + "YKM" + bases(2000) + "NNNN"
    rec2 = "NN" + bases(5000) + "N" * 130 + bases(4100) + "acgtnnnnACGT" + bases(77)
    rec3 = bases(1500) + "-" + bases(20) + "XX" + bases(901)
    with open(path, "w", newline="") as f:
        f.write(">chrA first record, with a comment\n")
        for o in range(0, len(rec1), 60):
            f.write(rec1[o:o + 60] + "\n")
        f.write("\n>chrB\tsecond\n")
        for o in range(0, len(rec2), 71):
            f.write(rec2[o:o + 71] + "\r\n")
        f.write(">chrC\n")
        for o in range(0, len(rec3), 50):
            f.write(rec3[o:o + 50] + "\n")


def odd_character_reads(genome, n101: int = 1400, n250: int = 500):
    """Reads for tests/golden/odd_characters.* (made by tests/golden/make_odd_characters.py with the reference's object code): spliced and indel reads of 101 and 250
    bases over `genome` (the pe101_spliced case's) with a literal '-' (a gap to AddNewCigarElements, tools.cpp:49-104), lower case, N and IUPAC letters at random
    places, next to junctions and inside stretches of noise between two seeds.  Single-end: every mate is a read of its own."""
    rng = np.random.default_rng(7711)
    seqs = []
    for rlen, n_pairs, seed in ((101, n101, 7701), (250, n250, 7702)):
        m1, m2 = synth.make_reads(genome, n_pairs, rlen=rlen, seed=seed, spliced_frac=0.6, indel_frac=0.3, n_frac=0.0)
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
                    s[int(rng.integers(0, rlen))] = ord("N"); s[int(rng.integers(0, rlen))] = ord("R")
                seqs.append(bytes(s))
    return seqs


def write_se_fastq(path: str, seqs) -> None:
    with open(path, "w") as f:
        for i, s in enumerate(seqs):
            f.write("@r%d\n%s\n+\n%s\n" % (i, s.decode(), "I" * len(s)))
