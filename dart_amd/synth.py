"""Seeded synthetic genomes and reads (SURVEY.md 8d "Synthetic inputs").

Real E. coli / GRCh38 cannot be fetched (no network), so every config of BASELINE.json is
run on an i.i.d. ACGT genome of the named size with planted repeat families, tandem repeats,
exact high-copy repeats (to exercise MaxDupNum, bwt_search.cpp:173) and planted introns with and
without GT..AG motifs (to exercise CheckSpliceJunction, AlignmentCandidates.cpp:758).

Reads: fragment length ~ N(350,40); mate 1 = fragment prefix, mate 2 = reverse complement of the
fragment suffix; strand flipped with p = 0.5; per-base substitutions; a fraction of pairs carries
one 1-3 bp indel; a fraction spans one planted intron; a fraction carries an 'N'.

Everything is a pure function of the seed (numpy PCG64), so the GPU box regenerates the same
bytes that the container used for its fixtures.
"""
from __future__ import annotations

import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
_COMP[:] = ord("N")
for _a, _b in zip(b"ACGTacgt", b"TGCATGCA"):
    _COMP[_a] = _b


def revcomp_ascii(a: np.ndarray) -> np.ndarray:
    return _COMP[a[::-1]]


class Genome:
    """codes: uint8 array of 0..3 over all chromosomes concatenated; offsets per chromosome."""

    def __init__(self, names, lengths, codes, introns):
        self.names = list(names)
        self.lengths = np.asarray(lengths, dtype=np.int64)
        self.offsets = np.concatenate([[0], np.cumsum(self.lengths)[:-1]]).astype(np.int64)
        self.codes = codes
        # introns: int64 [n,3] = (absolute start, length, has_motif) ; intron occupies [start, start+len)
        self.introns = introns

    @property
    def total(self) -> int:
        return int(self.lengths.sum())

    def ascii(self) -> np.ndarray:
        return _ACGT[self.codes]

    def write_fasta(self, path: str, width: int = 60) -> None:
        asc = self.ascii()
        with open(path, "wb") as f:
            for name, off, ln in zip(self.names, self.offsets, self.lengths):
                f.write(b">" + name.encode() + b"\n")
                seq = asc[off:off + ln]
                nfull = (ln // width) * width
                if nfull:
                    body = np.empty((nfull // width, width + 1), dtype=np.uint8)
                    body[:, :width] = seq[:nfull].reshape(-1, width)
                    body[:, width] = 10
                    f.write(body.tobytes())
                if ln > nfull:
                    f.write(seq[nfull:].tobytes() + b"\n")


def make_genome(lengths, seed=20, repeat_scale=1.0, n_introns=0, names=None) -> Genome:
    """i.i.d. genome + planted repeats. Repeat counts follow SURVEY 8d scaled by genome size
    relative to the chr20-sized config (64.4 Mbp): 30000 x 300 bp @10 %, 800 x 0.5-6 kb @5 %,
    plus exact high-copy and tandem repeats."""
    rng = np.random.default_rng(seed)
    lengths = [int(x) for x in lengths]
    total = sum(lengths)
    if names is None:
        names = ["chr%d" % (i + 1) for i in range(len(lengths))]
    codes = rng.integers(0, 4, size=total, dtype=np.uint8)
    scale = repeat_scale * total / 64444167.0

    def plant(n_copies, length, div, n_fam):
        for _ in range(n_fam):
            cons = rng.integers(0, 4, size=length, dtype=np.uint8)
            for _ in range(n_copies):
                p = int(rng.integers(0, max(1, total - length)))
                copy = cons.copy()
                if div > 0:
                    m = rng.random(length) < div
                    copy[m] = (copy[m] + rng.integers(1, 4, size=int(m.sum()), dtype=np.uint8)) & 3
                if rng.random() < 0.5:
                    copy = (3 - copy)[::-1]
                codes[p:p + length] = copy

    n_short = int(30000 * scale)
    fam = max(1, n_short // 60)
    plant(60, 300, 0.10, fam) if n_short >= 60 else plant(max(2, n_short), 300, 0.10, 1)
    n_long = max(2, int(800 * scale))
    for _ in range(max(1, n_long // 8)):
        plant(8, int(rng.integers(500, 6000)), 0.05, 1)
    # exact high-copy families: > MaxDupNum copies of the same 60-mer (freq > 100 -> no seed)
    plant(150, 60, 0.0, max(1, int(4 * scale)))
    # moderate exact families: 2..40 copies (multi-hit seeds, candidate ties)
    for c in (2, 3, 5, 12, 40):
        plant(c, 200, 0.0, max(1, int(20 * scale)))
    # tandem repeats: unit 11..47 bp, 4..12 copies
    for _ in range(max(2, int(400 * scale))):
        u = rng.integers(0, 4, size=int(rng.integers(11, 48)), dtype=np.uint8)
        k = int(rng.integers(4, 13))
        p = int(rng.integers(0, total - len(u) * k - 1))
        codes[p:p + len(u) * k] = np.tile(u, k)

    introns = np.zeros((0, 3), dtype=np.int64)
    if n_introns > 0:
        bounds = np.cumsum(lengths)
        starts = np.concatenate([[0], bounds[:-1]])
        rows = []
        tries = 0
        while len(rows) < n_introns and tries < n_introns * 20:
            tries += 1
            ci = int(rng.integers(0, len(lengths)))
            ilen = int(np.exp(rng.uniform(np.log(200), np.log(500000))))
            if lengths[ci] < ilen + 2000:
                continue
            s = int(starts[ci] + rng.integers(600, lengths[ci] - ilen - 600))
            motif = int(rng.random() < 0.5)
            if motif:
                if rng.random() < 0.5:  # GT..AG on the forward strand
                    codes[s:s + 2] = (2, 3)
                    codes[s + ilen - 2:s + ilen] = (0, 2)
                else:                    # CT..AC = GT..AG on the reverse strand
                    codes[s:s + 2] = (1, 3)
                    codes[s + ilen - 2:s + ilen] = (0, 1)
            rows.append((s, ilen, motif))
        introns = np.asarray(rows, dtype=np.int64).reshape(-1, 3)
    return Genome(names, lengths, codes, introns)


def make_reads(g: Genome, n_pairs: int, rlen: int = 101, seed: int = 7, sub_rate: float = 0.01,
               indel_frac: float = 0.02, spliced_frac: float = 0.0, n_frac: float = 0.002,
               paired: bool = True, frag_mean: float = 350.0, frag_sd: float = 40.0, return_truth: bool = False):
    """Returns (seq1, seq2) as uint8 [n_pairs, rlen] ASCII arrays (seq2 None when not paired).
    seq2 is the mate as sequenced (i.e. NOT yet reverse-complemented by the loader).
    return_truth: also a dict with, per pair, the chromosome index, the 1-based leftmost forward-strand position of each
    mate's footprint, and `plain` = the pair is a plain fragment (no planted indel / splice: its truth is exact)."""
    rng = np.random.default_rng(seed)
    asc = g.ascii()
    total = g.total
    ends = np.cumsum(g.lengths)
    flen = np.clip(np.rint(rng.normal(frag_mean, frag_sd, n_pairs)).astype(np.int64), rlen + 5, None)
    if not paired:
        flen[:] = rlen
    # choose start so the fragment stays inside one chromosome
    start = rng.integers(0, total, size=n_pairs).astype(np.int64)
    ci = np.searchsorted(ends, start, side="right")
    cend = ends[ci]
    start = np.where(start + flen + 8 > cend, np.maximum(g.offsets[ci], cend - flen - 8), start)
    flip = rng.random(n_pairs) < 0.5
    idx = start[:, None] + np.arange(rlen)[None, :]
    left = asc[idx]                                    # fragment prefix (forward)
    idx2 = (start + flen - rlen)[:, None] + np.arange(rlen)[None, :]
    right = asc[idx2]                                  # fragment suffix (forward)
    m1 = np.where(flip[:, None], _COMP[right[:, ::-1]], left)
    m2 = np.where(flip[:, None], left, _COMP[right[:, ::-1]])
    special = np.zeros(n_pairs, dtype=bool)
    # spliced / indel pairs are built one by one from an explicit "transcript"
    n_spl = int(round(spliced_frac * n_pairs)) if len(g.introns) else 0
    n_ind = int(round(indel_frac * n_pairs))
    which = rng.permutation(n_pairs)[: n_spl + n_ind]
    for k, p in enumerate(which):
        special[p] = True
        L = int(flen[p])
        if k < n_spl:
            s, ilen, _ = g.introns[int(rng.integers(0, len(g.introns)))]
            # put the junction inside mate 1's or mate 2's footprint, >= 20 bp from its ends
            off = int(rng.integers(20, rlen - 20))
            if rng.random() < 0.5:
                off = L - rlen + off
            a = int(s) - off
            lo = int(g.offsets[np.searchsorted(ends, s, side="right")])
            if a < lo:
                a = lo
                off = int(s) - a
            frag = np.concatenate([asc[a:a + off], asc[s + ilen:s + ilen + (L - off)]])
        else:
            a = int(start[p])
            frag = asc[a:a + L + 8].copy()
            pos = int(rng.integers(10, rlen - 10))
            if rng.random() < 0.5:
                pos = L - rlen + pos
            n = int(rng.integers(1, 4))
            if rng.random() < 0.5:
                frag = np.concatenate([frag[:pos], frag[pos + n:]])
            else:
                frag = np.concatenate([frag[:pos], _ACGT[rng.integers(0, 4, size=n)], frag[pos:]])
        if len(frag) < L:
            L = len(frag)
        if L < rlen:
            special[p] = False
            continue
        f1 = frag[:rlen]
        f2 = _COMP[frag[L - rlen:L][::-1]]
        if flip[p]:
            # fragment taken from the reverse strand: mate 1 = revcomp(suffix), mate 2 = prefix
            m1[p] = f2
            m2[p] = f1
        else:
            m1[p] = f1
            m2[p] = f2
    # substitutions
    for m in (m1, m2):
        mask = rng.random(m.shape) < sub_rate
        nsub = int(mask.sum())
        if nsub:
            cur = m[mask]
            code = np.zeros(256, dtype=np.uint8)
            code[ord("C")] = 1
            code[ord("G")] = 2
            code[ord("T")] = 3
            m[mask] = _ACGT[(code[cur] + rng.integers(1, 4, size=nsub, dtype=np.uint8)) & 3]
    # a few N
    n_n = int(round(n_frac * n_pairs))
    if n_n:
        rows = rng.integers(0, n_pairs, size=n_n)
        cols = rng.integers(0, rlen, size=n_n)
        m1[rows, cols] = ord("N")
        rows = rng.integers(0, n_pairs, size=n_n)
        cols = rng.integers(0, rlen, size=n_n)
        m2[rows, cols] = ord("N")
    if return_truth:
        lo1 = np.where(flip, start + flen - rlen, start)
        lo2 = np.where(flip, start, start + flen - rlen)
        truth = {"chr": ci.astype(np.int32), "pos1": (lo1 - g.offsets[ci] + 1).astype(np.int64),
                 "pos2": (lo2 - g.offsets[ci] + 1).astype(np.int64), "plain": ~special}
        return (m1, m2, truth) if paired else (m1, None, truth)
    return (m1, m2) if paired else (m1, None)


def write_fastq(path: str, seqs: np.ndarray, mate: int, qual: int = ord("I")) -> None:
    n, rlen = seqs.shape
    ids = np.char.add(np.char.add("@r", np.arange(n).astype(str)), "/%d\n" % mate)
    q = bytes([qual]) * rlen
    with open(path, "wb") as f:
        chunk = []
        for i in range(n):
            chunk.append(ids[i].encode())
            chunk.append(seqs[i].tobytes())
            chunk.append(b"\n+\n")
            chunk.append(q)
            chunk.append(b"\n")
            if len(chunk) > 50000:
                f.write(b"".join(chunk))
                chunk = []
        f.write(b"".join(chunk))


def write_fasta_reads(path: str, seqs: np.ndarray, mate: int) -> None:
    with open(path, "wb") as f:
        for i in range(seqs.shape[0]):
            f.write(b">r%d/%d\n" % (i, mate))
            f.write(seqs[i].tobytes())
            f.write(b"\n")
