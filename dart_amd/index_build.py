"""BWA-format index builder (PREFIX.pac/.ann/.amb/.bwt/.sa) -- replaces the *output* of the
reference's offline indexer (`bwt_index`, BWT_Index/bwtindex.c:77-148) for the synthetic genomes
the tests and bench.py generate on the GPU box, where no reference binary exists.

The index files are a pure function of the text, so any correct suffix sorter reproduces the
reference's files byte for byte ("Format validated here"); tests/test_index_build.py
checks that against oracle/_ref/bwt_index.  The suffix array comes from prefix doubling with
torch.sort, which runs on the MI355X when one is present (chr20-sized text: seconds) and on the
CPU otherwise (fine for the <= few-Mbp test genomes).  This is scope row 8f#1 ("next"), kept in
Python on purpose: it is plumbing around the hot path, not the hot path.

Layout facts restated from the reference:
  .pac  forward strand, 2 bit/base MSB first; +1 zero byte when l_pac%4==0; last byte = l_pac%4
        (bntseq.c:192-201).  N -> lrand48()&3 after srand48(11) (bntseq.c:144,173-174).
  .bwt  primary, L2[1..4] (u64) then per 128 symbols 4 x u64 counts + 8 x u32 of 16 symbols, and
        a final 4 x u64 (bwtindex.c:53-75).  `$` is removed; text = forward + reverse complement.
  .sa   primary, L2[1..4], 32, seq_len (u64) then SA[32], SA[64], ... (bwt.c:101-123,185-196).
"""
from __future__ import annotations

import numpy as np
import torch

_NT4 = np.full(256, 4, dtype=np.uint8)
for _i, _c in enumerate(b"ACGT"):
    _NT4[_c] = _i
    _NT4[_c + 32] = _i


def read_fasta(path: str):
    names, annos, seqs = [], [], []
    cur = []
    opener = open
    if path.endswith(".gz"):
        import gzip
        opener = gzip.open
    with opener(path, "rb") as f:
        for line in f:
            if line.startswith(b">"):
                if names:
                    seqs.append(b"".join(cur))
                cur = []
                parts = line[1:].strip().split(None, 1)
                names.append(parts[0].decode() if parts else "")
                annos.append(parts[1].decode() if len(parts) > 1 else "")
            else:
                cur.append(line.strip())
    if names:
        seqs.append(b"".join(cur))
    return names, annos, seqs


class _LRand48:
    """glibc srand48/lrand48."""

    def __init__(self, seed: int):
        self.x = ((seed & 0xFFFFFFFF) << 16) | 0x330E

    def next(self) -> int:
        self.x = (self.x * 0x5DEECE66D + 0xB) & ((1 << 48) - 1)
        return self.x >> 17


def pack_sequences(seqs):
    """-> (codes uint8 0..3 of all sequences concatenated, holes list, n_ambs per sequence)."""
    rng = _LRand48(11)
    out = []
    holes = []
    n_ambs = []
    off = 0
    for s in seqs:
        a = np.frombuffer(s, dtype=np.uint8)
        c = _NT4[a].copy()
        bad = np.nonzero(c >= 4)[0]
        na = 0
        if len(bad):
            last_pos, last_ch = -2, -1
            for p in bad:               # bntseq.c:125-146: a hole = run of the same ambiguous char
                ch = int(a[p])
                if p == last_pos + 1 and ch == last_ch:
                    holes[-1][1] += 1
                else:
                    holes.append([off + int(p), 1, chr(ch)])
                    na += 1
                last_pos, last_ch = int(p), ch
                c[p] = rng.next() & 3
        n_ambs.append(na)
        out.append(c)
        off += len(a)
    return (np.concatenate(out) if out else np.zeros(0, np.uint8)), holes, n_ambs


def suffix_array(codes: torch.Tensor) -> torch.Tensor:
    """Suffix array of codes+'$' ('$' smallest) by prefix doubling. Returns int64 [n+1]."""
    dev = codes.device
    n = int(codes.numel())
    N = n + 1
    t = torch.zeros(N + 32, dtype=torch.int64, device=dev)
    t[:n] = codes.to(torch.int64) + 1
    k0 = 16
    key = torch.zeros(N, dtype=torch.int64, device=dev)
    for j in range(k0):
        key = key * 5 + t[j:j + N]
    del t
    sk, sa = torch.sort(key)
    del key
    flag = torch.ones(N, dtype=torch.int64, device=dev)
    flag[1:] = (sk[1:] != sk[:-1]).to(torch.int64)
    flag[0] = 0
    del sk
    rs = torch.cumsum(flag, 0)
    rank = torch.empty(N, dtype=torch.int64, device=dev)
    rank[sa] = rs
    k = k0
    while int(rs[-1]) < N - 1:
        r2 = torch.zeros(N, dtype=torch.int64, device=dev)
        if k < N:
            r2[:N - k] = rank[k:] + 1
        key = rank * (N + 1) + r2
        del r2
        sk, sa = torch.sort(key)
        del key
        flag = torch.ones(N, dtype=torch.int64, device=dev)
        flag[1:] = (sk[1:] != sk[:-1]).to(torch.int64)
        flag[0] = 0
        del sk
        rs = torch.cumsum(flag, 0)
        rank[sa] = rs
        k *= 2
    return sa


def build_index(prefix: str, names, annos, seqs, device: str | None = None) -> dict:
    if device is None:
        device = "cuda" if torch.cuda.is_available() else "cpu"
    fwd, holes, n_ambs = pack_sequences(seqs)
    L = int(len(fwd))
    # ---- .pac / .ann / .amb ----
    pad = np.zeros((-L) % 4, dtype=np.uint8)
    f4 = np.concatenate([fwd, pad]).reshape(-1, 4)
    pac = ((f4[:, 0] << 6) | (f4[:, 1] << 4) | (f4[:, 2] << 2) | f4[:, 3]).astype(np.uint8)
    with open(prefix + ".pac", "wb") as f:
        f.write(pac.tobytes())
        if L % 4 == 0:
            f.write(b"\0")
        f.write(bytes([L % 4]))
    with open(prefix + ".ann", "w") as f:
        f.write("%d %d %u\n" % (L, len(seqs), 11))
        off = 0
        for name, anno, s, na in zip(names, annos, seqs, n_ambs):
            f.write("0 %s %s\n" % (name, anno if anno else "(null)"))
            f.write("%d %d %d\n" % (off, len(s), na))
            off += len(s)
    with open(prefix + ".amb", "w") as f:
        f.write("%d %d %u\n" % (L, len(seqs), len(holes)))
        for o, ln, ch in holes:
            f.write("%d %d %s\n" % (o, ln, ch))
    # ---- text = forward + reverse complement ----
    text = np.concatenate([fwd, (3 - fwd)[::-1]])
    n = 2 * L
    sa = suffix_array(torch.from_numpy(text).to(device))
    primary = int(torch.nonzero(sa == 0)[0, 0])
    tt = torch.from_numpy(text).to(device)
    prev = tt[(sa - 1).clamp(min=0)]
    keep = torch.ones(n + 1, dtype=torch.bool, device=sa.device)
    keep[primary] = False
    bwt = prev[keep].cpu().numpy().astype(np.uint8)           # n symbols, '$' row removed
    sa_s = sa[32::32].cpu().numpy().astype(np.uint64)         # rows 32, 64, ... of the (n+1)-row matrix
    del sa, prev, keep, tt
    cnt = np.bincount(text, minlength=4).astype(np.uint64)
    L2 = np.concatenate([[0], np.cumsum(cnt)]).astype(np.uint64)
    # ---- Occ-interleaved .bwt ----
    nblk = (n + 127) // 128
    bp = np.zeros(nblk * 128, dtype=np.uint8)
    bp[:n] = bwt
    w16 = bp.reshape(-1, 16).astype(np.uint32)
    shifts = (30 - 2 * np.arange(16)).astype(np.uint32)
    words = (w16 << shifts).sum(axis=1, dtype=np.uint64).astype(np.uint32).reshape(nblk, 8)
    valid = np.zeros(nblk * 128, dtype=bool)
    valid[:n] = True
    occ = np.zeros((nblk + 1, 4), dtype=np.uint64)
    b128 = bp.reshape(nblk, 128)
    v128 = valid.reshape(nblk, 128)
    for c in range(4):
        occ[1:, c] = np.cumsum(((b128 == c) & v128).sum(axis=1)).astype(np.uint64)
    blocks = np.zeros((nblk, 16), dtype=np.uint32)
    blocks[:, :8] = occ[:-1].view(np.uint32).reshape(nblk, 8)
    blocks[:, 8:] = words
    flat = blocks.reshape(-1)
    nwords = (n + 15) // 16
    body = flat[: (nblk - 1) * 16 + 8 + (nwords - (nblk - 1) * 8)] if nblk else flat[:0]
    with open(prefix + ".bwt", "wb") as f:
        f.write(np.array([primary], dtype=np.uint64).tobytes())
        f.write(L2[1:5].tobytes())
        f.write(body.tobytes())
        f.write(occ[-1].tobytes())
    with open(prefix + ".sa", "wb") as f:
        f.write(np.array([primary], dtype=np.uint64).tobytes())
        f.write(L2[1:5].tobytes())
        f.write(np.array([32, n], dtype=np.uint64).tobytes())
        f.write(sa_s[: (n + 32) // 32 - 1].tobytes())
    return {"l_pac": L, "seq_len": n, "primary": primary}


def build_index_from_fasta(fasta: str, prefix: str, device: str | None = None) -> dict:
    names, annos, seqs = read_fasta(fasta)
    return build_index(prefix, names, annos, seqs, device)


def build_index_from_genome(g, prefix: str, device: str | None = None) -> dict:
    """g: dart_amd.synth.Genome (no FASTA round trip)."""
    asc = g.ascii()
    seqs = [asc[o:o + l].tobytes() for o, l in zip(g.offsets, g.lengths)]
    return build_index(prefix, g.names, [""] * len(seqs), seqs, device)
