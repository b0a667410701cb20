"""BWA-format index builder (PREFIX.pac/.ann/.amb/.bwt/.sa) -- replaces the *output* of the
reference's offline indexer (`bwt_index`, BWT_Index/bwtindex.c:77-148) for the synthetic genomes
the tests and bench.py generate on the GPU box, where no reference binary exists.

The index files are a pure function of the text, so any correct suffix sorter reproduces the
reference's files byte for byte ("Format validated here");
tests/test_oracle_golden.py::test_index_builder_matches_reference_indexer checks that against the digests of
oracle/_ref/bwt_index's files (CPU path) and tests/test_gpu_index.py does the same for the GPU path.  The suffix array comes from prefix doubling with
one sort of (rank pair, suffix) per round: on the MI355X that sort is the library's own radix sort (dg_sort_pairs,
dart_amd/csrc/dg_sort.h -- hand-written HIP, 4 bits per pass, LDS-staged tiles), on the CPU (the <= few-Mbp genomes of the CPU
suite) torch.sort.  The orchestration around the sort (rank updates, bucketing, BWT/Occ packing) is element-wise torch: plumbing.
Scope row 8f#1 ("next").

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


_lib = None
_sort_s = [0.0, 0, 0]                                   # seconds inside the sorter, calls, pairs (for the phase log)


def sort_pairs(key: torch.Tensor, key_bits: int):
    """(sorted keys, order) of a non-negative int64 key tensor.  On the GPU this is the library's own radix sort (dg_sort_pairs,
    dart_amd/csrc/dg_sort.h: stable, 4 bits per pass, key_bits says how many low bits matter); on the CPU (the small test genomes of
    the CPU suite) torch.sort.  No silent fallback on the GPU: without libdartgpu.so the build fails."""
    if key.device.type != "cuda":
        return torch.sort(key)
    global _lib
    if _lib is None:
        import ctypes as C
        from . import host
        _lib = host._load_lib()
        _lib.dg_sort_pairs.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    m = int(key.numel())
    vals = torch.arange(m, dtype=torch.int64, device=key.device)
    if m < 2:
        return key, vals
    key = key.contiguous()
    tk, tv = torch.empty_like(key), torch.empty_like(vals)
    torch.cuda.current_stream(key.device).synchronize()          # (the sorter runs on the NULL stream)
    import time as _time
    t0 = _time.time()
    rc = _lib.dg_sort_pairs(key.device.index or 0, key.data_ptr(), vals.data_ptr(), tk.data_ptr(), tv.data_ptr(), m, int(key_bits))
    _sort_s[0] += _time.time() - t0; _sort_s[1] += 1; _sort_s[2] += m
    if rc != 0:
        raise RuntimeError("dg_sort_pairs failed (%d)" % rc)
    return key, vals


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
    sk, sa = sort_pairs(key, 38)                       # 16 symbols base 5 < 2^38
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
        sk, sa = sort_pairs(key, max(1, (N * (N + 1) + N).bit_length()))
        del key
        flag = torch.ones(N, dtype=torch.int64, device=dev)
        flag[1:] = (sk[1:] != sk[:-1]).to(torch.int64)
        flag[0] = 0
        del sk
        rs = torch.cumsum(flag, 0)
        rank[sa] = rs
        k *= 2
    return sa


def suffix_array_bucketed(codes: torch.Tensor, log=None) -> torch.Tensor:
    """Suffix array of codes+'$' for texts too large for `suffix_array` (which keeps ~9 arrays of N int64 alive:
    150 GB at 2 G symbols): the suffixes are split by their first two symbols into <= 21 buckets whose relative order
    is known, and prefix doubling runs bucket by bucket -- the sort temporaries are bucket-sized, only sa, rank and the
    next round's rank are full-length (3 x 8 N bytes: 150 GB at GRCh38 size, N = 6.2 G).  After the first round only
    the members of groups that are still tied are sorted again (Larsson-Sadakane style), so later rounds cost what the
    repeats cost.  Same result as `suffix_array` (the suffix array is unique)."""
    dev = codes.device
    n = int(codes.numel())
    N = n + 1
    k0 = 16
    t = torch.zeros(N + k0 + 1, dtype=torch.uint8, device=dev)
    t[:n] = codes + 1                                            # '$' (and the padding behind it) = 0
    CHK = 1 << 30                                                 # torch.nonzero / bincount are limited to < 2^31 elements
    bid = torch.empty(N, dtype=torch.uint8, device=dev)
    counts = torch.zeros(25, dtype=torch.int64)
    for c0 in range(0, N, CHK):
        c1 = min(N, c0 + CHK)
        bid[c0:c1] = (t[c0:c1].to(torch.int16) * 5 + t[c0 + 1:c1 + 1].to(torch.int16)).to(torch.uint8)
        counts += torch.bincount(bid[c0:c1].to(torch.int32), minlength=25).to(torch.int64).cpu()
    offs = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(counts, 0)])
    assert int(counts.max()) < (1 << 29), "a two-symbol bucket holds >= 2^29 suffixes: use three-symbol buckets"
    sa = torch.empty(N, dtype=torch.int64, device=dev)
    rank = torch.empty(N, dtype=torch.int64, device=dev)
    pending = {}                                                  # bucket -> positions (in the bucket) still in tied groups

    def regroup(sk, lo, members, pos_in_bucket, rank_out):
        """sk: sorted keys of the processed elements (in order), members: their suffixes in that order,
        pos_in_bucket: their (ascending) positions.  Writes ranks = absolute position of each group's first element;
        returns the positions that are still tied."""
        m = sk.numel()
        flag = torch.ones(m, dtype=torch.bool, device=dev)
        if m > 1:
            flag[1:] = sk[1:] != sk[:-1]
        start = torch.cummax(torch.where(flag, pos_in_bucket, torch.zeros_like(pos_in_bucket)), 0).values
        rank_out[members] = start + lo
        single = flag.clone()
        if m > 1:
            single[:-1] &= flag[1:]
        return pos_in_bucket[~single]

    for b in range(25):                                           # round 0: the first k0 symbols
        m = int(counts[b])
        if m == 0:
            continue
        lo = int(offs[b])
        idx = torch.cat([torch.nonzero(bid[c0:min(N, c0 + CHK)] == b).squeeze(1) + c0 for c0 in range(0, N, CHK)])
        key = torch.zeros(m, dtype=torch.int64, device=dev)
        for j in range(k0):
            key = key * 5 + t[idx + j].to(torch.int64)
        sk, order = sort_pairs(key, 38)
        del key
        members = idx[order]
        del idx, order
        sa[lo:lo + m] = members
        rest = regroup(sk, lo, members, torch.arange(m, dtype=torch.int64, device=dev), rank)
        del sk, members
        if rest.numel():
            pending[b] = rest
        if log:
            log("  bucket %2d: %d suffixes, %d still tied after %d symbols" % (b, m, int(rest.numel()), k0))
    del bid
    k = k0
    while pending:
        rank_new = rank.clone()
        for b in sorted(pending):
            lo = int(offs[b])
            pos = pending[b]                                      # ascending positions inside the bucket
            members = sa[lo + pos]
            r1 = rank[members] - lo                               # < 2^29
            nxt = members + k
            r2 = torch.where(nxt < N, rank[nxt.clamp(max=N - 1)] + 1, torch.zeros_like(nxt))
            key = (r1 << 34) | r2                                 # r2 <= N < 2^34
            del r1, r2, nxt
            sk, order = sort_pairs(key, 63)
            del key
            members = members[order]
            del order
            sa[lo + pos] = members                                # a tied group occupies the same positions before and after
            rest = regroup(sk, lo, members, pos, rank_new)
            del sk, members
            if rest.numel():
                pending[b] = rest
            else:
                del pending[b]
        rank = rank_new
        del rank_new
        k *= 2
        if log:
            log("  after %d symbols: %d suffixes still tied" % (k, sum(int(v.numel()) for v in pending.values())))
    return sa


# ------------------------------------------------------------------------------------------------------------------------------
# The MI355X path: every per-symbol step is a HIP kernel of libdartindex.so (dart_amd/csrc/index/dg_index.hip, C ABI
# include/dartindex.h) or the radix sorter of libdartgpu.so; torch allocates the device memory and does the per-bucket bookkeeping
# (prefix sums over per-tile counts, the cumulative Occ counts over 48 M blocks).  No fallback: on a GPU box a missing library is an error.
_ilib = None


def _index_lib():
    global _ilib
    if _ilib is None:
        import ctypes as C, os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdartindex.so")
        if not os.path.exists(path):
            raise RuntimeError("libdartindex.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'`); the GPU index builder has no CPU fallback")
        lib = C.CDLL(path)
        vp, u64, u32 = C.c_void_p, C.c_uint64, C.c_uint32
        lib.di_last_error.restype = C.c_char_p
        lib.di_text_words.restype = C.c_size_t
        lib.di_text_words.argtypes = [u64]
        lib.di_pack_text.argtypes = [C.c_int, vp, u64, vp]
        lib.di_bucket_hist.argtypes = [C.c_int, vp, u64, vp]
        lib.di_bucket_keys.argtypes = [C.c_int, vp, u64, C.c_int, vp, vp, vp]
        lib.di_doubling_keys.argtypes = [C.c_int, vp, vp, u64, vp, u32, u64, u64, C.c_int, vp, vp]
        lib.di_regroup.argtypes = [C.c_int, vp, vp, vp, u32, u64, vp, vp, vp, vp, C.POINTER(u32)]
        lib.di_bwt_blocks.argtypes = [C.c_int, vp, vp, u64, u64, vp, vp]
        lib.DI_LOG = C.CFUNCTYPE(None, C.c_char_p, vp)
        lib.di_build_files.argtypes = [C.c_int, vp, u64, C.c_char_p, lib.DI_LOG, vp, C.POINTER(u64)]
        _ilib = lib
    return _ilib


def _di(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed (%d): %s" % (what, rc, (_index_lib().di_last_error() or b"").decode()))


def _sort_kv(keys, vals, tk, tv, m, key_bits):
    """dg_sort_pairs on the first m pairs of (keys, vals); tk / tv are scratch of at least m elements."""
    import ctypes as C, time as _time
    global _lib
    if _lib is None:
        from . import host
        _lib = host._load_lib()
        _lib.dg_sort_pairs.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    if m < 2:
        return
    torch.cuda.current_stream(keys.device).synchronize()
    t0 = _time.time()
    rc = _lib.dg_sort_pairs(keys.device.index or 0, keys.data_ptr(), vals.data_ptr(), tk.data_ptr(), tv.data_ptr(), m, int(key_bits))
    _sort_s[0] += _time.time() - t0; _sort_s[1] += 1; _sort_s[2] += m
    if rc != 0:
        raise RuntimeError("dg_sort_pairs failed (%d)" % rc)


DI_TILE = 4096


class _HipOps:
    """The device steps of suffix_array_hip, one method per C entry point of include/dartindex.h (tensors in, nothing copied).  The CPU
    suite swaps in tests/index_emul.py's numpy restatement of the same contracts to check the scheme and this file's bookkeeping; the
    kernels themselves are checked on the GPU (tests/test_gpu_index.py: the reference indexer's bytes)."""

    def __init__(self, dev):
        self.lib, self.dev, self.di = _index_lib(), dev, dev.index or 0

    def sync(self):
        torch.cuda.synchronize(self.dev)

    def bucket_hist(self, T, n, table):
        self.sync(); _di(self.lib.di_bucket_hist(self.di, T.data_ptr(), n, table.data_ptr()), "di_bucket_hist")

    def bucket_keys(self, T, n, pair, base, keys, vals):
        self.sync(); _di(self.lib.di_bucket_keys(self.di, T.data_ptr(), n, pair, base.data_ptr(), keys.data_ptr(), vals.data_ptr()), "di_bucket_keys")

    def doubling_keys(self, sa, rank, lo, pos, m, k, N, r2_bits, keys, vals):
        self.sync(); _di(self.lib.di_doubling_keys(self.di, sa.data_ptr(), rank.data_ptr(), lo, pos.data_ptr(), m, k, N, r2_bits, keys.data_ptr(), vals.data_ptr()), "di_doubling_keys")

    def sort(self, keys, vals, tk, tv, m, key_bits):
        _sort_kv(keys, vals, tk, tv, m, key_bits)

    def regroup(self, keys, vals, pos, m, lo, rank, sa, new_pos, scratch) -> int:
        import ctypes as C
        n_tied = C.c_uint32(0)
        self.sync()
        _di(self.lib.di_regroup(self.di, keys.data_ptr(), vals.data_ptr(), pos.data_ptr() if pos is not None else None, m, lo, rank.data_ptr(), sa.data_ptr(),
                                new_pos.data_ptr(), scratch.data_ptr(), C.byref(n_tied)), "di_regroup")
        return int(n_tied.value)


def suffix_array_hip(T: torch.Tensor, n: int, last_sym: int, log=None, ops=None):
    """Suffix array AND inverse (sa, rank: int64 [n + 1]) of the packed text T (di_pack_text's layout) + '$': bucketed prefix doubling with
    every per-suffix step on the device (the scheme is described at the top of dg_index.hip).  last_sym = the text's last symbol."""
    dev = T.device
    ops = ops or _HipOps(dev)
    N = n + 1
    assert n >= 64
    tiles = (N + DI_TILE - 1) // DI_TILE
    table = torch.empty(16 * tiles, dtype=torch.int32, device=dev)
    ops.bucket_hist(T, n, table)
    table = table.view(16, tiles)
    counts = [int(x) for x in table.sum(dim=1, dtype=torch.int64).cpu()]
    assert sum(counts) == n - 1
    m_max = max(counts)
    assert m_max < (1 << 31) - DI_TILE, "a two-symbol bucket holds >= 2^31 suffixes"
    r2_bits = N.bit_length()                                      # a rank + 1 is at most N
    assert r2_bits + m_max.bit_length() <= 64, "rank pair does not fit 64 bits"
    sa = torch.empty(N, dtype=torch.int64, device=dev)
    rank = torch.empty(N, dtype=torch.int64, device=dev)
    keys = torch.empty(m_max, dtype=torch.int64, device=dev)
    vals = torch.empty(m_max, dtype=torch.int64, device=dev)
    tk, tv = torch.empty_like(keys), torch.empty_like(vals)
    new_pos = torch.empty(m_max, dtype=torch.int32, device=dev)
    scratch = torch.empty(2 * ((m_max + DI_TILE - 1) // DI_TILE) + 4, dtype=torch.int32, device=dev)
    # rows: '$' first, then per first symbol c0: the suffix "c0 $" (if the text ends in c0), then the buckets c0 A, c0 C, c0 G, c0 T
    sa[0] = n
    rank[n] = 0
    row = 1
    lows = {}
    for c0 in range(4):
        if c0 == last_sym:
            sa[row] = n - 1
            rank[n - 1] = row
            row += 1
        for c1 in range(4):
            lows[c0 * 4 + c1] = row
            row += counts[c0 * 4 + c1]
    assert row == N
    pending = {}
    for pair in range(16):                                        # round 0: the first 31 symbols
        m, lo = counts[pair], lows[pair]
        if m == 0:
            continue
        r = table[pair]
        base = (torch.cumsum(r, 0) - r).to(torch.int32)
        ops.bucket_keys(T, n, pair, base, keys, vals)
        del base
        ops.sort(keys, vals, tk, tv, m, 63)
        t = ops.regroup(keys, vals, None, m, lo, rank, sa, new_pos, scratch)
        if t:
            pending[pair] = new_pos[:t].clone()
        if log:
            log("  bucket %s: %d suffixes, %d still tied after 31 symbols" % ("ACGT"[pair >> 2] + "ACGT"[pair & 3], m, t))
    del table
    k = 31
    while pending:
        for pair in sorted(pending):
            pos, lo = pending[pair], lows[pair]
            m = int(pos.numel())
            ops.doubling_keys(sa, rank, lo, pos, m, k, N, r2_bits, keys, vals)
            ops.sort(keys, vals, tk, tv, m, min(64, r2_bits + counts[pair].bit_length()))
            t = ops.regroup(keys, vals, pos, m, lo, rank, sa, new_pos, scratch)
            if t:
                pending[pair] = new_pos[:t].clone()
            else:
                del pending[pair]
        k *= 2
        if log:
            log("  after %d symbols: %d suffixes still tied" % (k, sum(int(v.numel()) for v in pending.values())))
    return sa, rank


def _build_files_hip(prefix: str, fwd: np.ndarray, device: str, log=None) -> dict:
    """.pac, .bwt, .sa of the forward codes `fwd` (uint8 0..3) on the MI355X: di_build_files, the library's own driver (the one `dart index`
    uses).  DART_INDEX_DRIVER=python runs the same kernels from this file instead (suffix_array_hip below): a cross-check of the two drivers,
    and the form whose bookkeeping the CPU suite can follow (tests/test_index_scheme.py)."""
    import ctypes as C, os as _os
    lib = _index_lib()
    dev = torch.device(device if ":" in device else device + ":0")
    di = dev.index or 0
    L = int(len(fwd))
    n = 2 * L
    if _os.environ.get("DART_INDEX_DRIVER", "") != "python":
        fwd = np.ascontiguousarray(fwd, dtype=np.uint8)
        cb = lib.DI_LOG((lambda line, arg: log(line.decode())) if log else (lambda line, arg: None))
        primary = C.c_uint64(0)
        torch.cuda.synchronize(dev)
        _di(lib.di_build_files(di, fwd.ctypes.data, L, prefix.encode(), cb, None, C.byref(primary)), "di_build_files")
        return {"l_pac": L, "seq_len": n, "primary": int(primary.value)}
    f = torch.from_numpy(fwd).to(dev)
    if log: log("  forward codes on the device")
    # .pac: 4 symbols per byte, first symbol in the top bits (bntseq.c:192-201)
    f4 = torch.zeros((L + 3) // 4 * 4, dtype=torch.uint8, device=dev)
    f4[:L] = f
    f4 = f4.view(-1, 4)
    pac = (f4[:, 0] << 6) | (f4[:, 1] << 4) | (f4[:, 2] << 2) | f4[:, 3]
    pac_h = pac.cpu().numpy()
    del f4, pac
    with open(prefix + ".pac", "wb") as fh:
        pac_h.tofile(fh)
        if L % 4 == 0:
            fh.write(b"\0")
        fh.write(bytes([L % 4]))
    del pac_h
    if log: log("  .pac written")
    T = torch.empty(int(lib.di_text_words(n)), dtype=torch.int64, device=dev)
    torch.cuda.synchronize(dev)
    _di(lib.di_pack_text(di, f.data_ptr(), L, T.data_ptr()), "di_pack_text")
    last_sym = 3 - int(fwd[0])
    del f
    if log: log("  text packed (forward + reverse complement, 2 bits per symbol)")
    sa, rank = suffix_array_hip(T, n, last_sym, log)
    primary = int(rank[0])                                        # the row of suffix 0
    del rank
    if log: log("  suffix array done, primary = %d" % primary)
    sa_s = sa[32::32].cpu().numpy().view(np.uint64)               # rows 32, 64, ... of the (n+1)-row matrix
    if log: log("  SA sampled")
    nblk = (n + 127) // 128
    blocks = torch.zeros(nblk * 16, dtype=torch.int32, device=dev)
    c4 = torch.empty(nblk, dtype=torch.int32, device=dev)
    torch.cuda.synchronize(dev)
    _di(lib.di_bwt_blocks(di, sa.data_ptr(), T.data_ptr(), n, primary, blocks.data_ptr(), c4.data_ptr()), "di_bwt_blocks")
    del sa, T
    if log: log("  BWT gathered (16 symbols per word, symbol counts per block)")
    per = torch.stack([(c4 >> (8 * c)) & 255 for c in range(4)], dim=0).to(torch.int64)      # [4, blocks] symbols per block; one contiguous row per symbol
    del c4
    cum = torch.cumsum(per, 1)
    blocks.view(torch.int64).view(nblk, 8)[:, :4] = (cum - per).t()                          # the counts in front of each block
    occ_last = cum[:, -1].cpu().numpy().astype(np.uint64)
    del per, cum
    if log: log("  Occ counts in the blocks")
    flat = blocks.cpu().numpy().view(np.uint32)
    del blocks
    if log: log("  blocks on the host")
    L2 = np.concatenate([[0], np.cumsum(occ_last)]).astype(np.uint64)          # the BWT is a permutation of the text: its totals are the symbol counts
    nwords = (n + 15) // 16
    body = flat[: (nblk - 1) * 16 + 8 + (nwords - (nblk - 1) * 8)]
    with open(prefix + ".bwt", "wb") as fh:
        fh.write(np.array([primary], dtype=np.uint64).tobytes())
        fh.write(L2[1:5].tobytes())
        body.tofile(fh)
        fh.write(occ_last.tobytes())
    if log: log("  .bwt written")
    with open(prefix + ".sa", "wb") as fh:
        fh.write(np.array([primary], dtype=np.uint64).tobytes())
        fh.write(L2[1:5].tobytes())
        fh.write(np.array([32, n], dtype=np.uint64).tobytes())
        sa_s[: (n + 32) // 32 - 1].tofile(fh)
    if log: log("  .sa written")
    return {"l_pac": L, "seq_len": n, "primary": primary}


def _pack_bwt_occ(bwt: torch.Tensor, n: int):
    """bwt: uint8 [n] on any device -> (blocks uint32 [nblk,16] numpy, occ_last uint64 [4]) in the .bwt layout, in chunks."""
    dev = bwt.device
    nblk = (n + 127) // 128
    blocks = np.zeros((nblk, 16), dtype=np.uint32)
    shifts = (30 - 2 * torch.arange(16, device=dev)).to(torch.int64)
    run = torch.zeros(4, dtype=torch.int64, device=dev)
    CH = 1 << 20                                                  # blocks per chunk (128 M symbols)
    for b0 in range(0, nblk, CH):
        b1 = min(nblk, b0 + CH)
        seg = torch.zeros((b1 - b0) * 128, dtype=torch.uint8, device=dev)
        s0, s1 = b0 * 128, min(n, b1 * 128)
        seg[: s1 - s0] = bwt[s0:s1]
        valid = torch.zeros((b1 - b0) * 128, dtype=torch.bool, device=dev)
        valid[: s1 - s0] = True
        words = (seg.view(-1, 16).to(torch.int64) << shifts).sum(dim=1).view(b1 - b0, 8)
        b128, v128 = seg.view(b1 - b0, 128), valid.view(b1 - b0, 128)
        per = torch.stack([((b128 == c) & v128).sum(dim=1) for c in range(4)], dim=1).to(torch.int64)   # [blocks,4]
        cum = torch.cumsum(per, 0) + run                           # counts up to and including each block
        before = cum - per
        run = cum[-1].clone()
        blocks[b0:b1, :8] = before.cpu().numpy().astype(np.uint64).view(np.uint32).reshape(b1 - b0, 8)
        blocks[b0:b1, 8:] = words.cpu().numpy().astype(np.uint32)
        del seg, valid, words, per, cum, before
    return blocks, run.cpu().numpy().astype(np.uint64)


def build_index(prefix: str, names, annos, seqs, device: str | None = None, log=None, codes=None) -> dict:
    """seqs: the sequences as bytes (FASTA text), or -- with codes = their concatenated 0..3 codes, no ambiguous base -- just their lengths.
    device "cuda": the HIP builder (DART_SA_TORCH=plain|bucketed selects the older torch-orchestrated sorters, kept as cross-checks);
    "cpu": torch on the host, for the small genomes of the CPU suite."""
    if device is None:
        device = "cuda" if torch.cuda.is_available() else "cpu"
    if log:                                                       # every log line carries the seconds since the build began
        import time as _time
        _t0, _log0 = _time.time(), log
        def log(msg):
            if device != "cpu": torch.cuda.synchronize()
            if msg.lstrip().startswith("["):                     # di_build_files' own lines carry their clock already
                _log0(msg)
            else:
                _log0("  [%6.1f s; sorter %5.1f s, %d calls, %.2f G pairs]%s" % (_time.time() - _t0, _sort_s[0], _sort_s[1], _sort_s[2] / 1e9, msg))
    import os as _os
    if codes is None:
        fwd, holes, n_ambs = pack_sequences(seqs)
        lengths = [len(s) for s in seqs]
    else:
        fwd, holes, n_ambs, lengths = np.ascontiguousarray(codes, dtype=np.uint8), [], [0] * len(seqs), [int(x) for x in seqs]
        assert sum(lengths) == len(fwd)
    if log: log("  sequences packed")
    L = int(len(fwd))
    # ---- .ann / .amb ----
    with open(prefix + ".ann", "w") as f:
        f.write("%d %d %u\n" % (L, len(seqs), 11))
        off = 0
        for name, anno, ln, na in zip(names, annos, lengths, n_ambs):
            f.write("0 %s %s\n" % (name, anno if anno else "(null)"))
            f.write("%d %d %d\n" % (off, ln, na))
            off += ln
    with open(prefix + ".amb", "w") as f:
        f.write("%d %d %u\n" % (L, len(seqs), len(holes)))
        for o, ln, ch in holes:
            f.write("%d %d %s\n" % (o, ln, ch))
    # ---- text = forward + reverse complement ----
    torch_sorter = _os.environ.get("DART_SA_TORCH", "")
    if str(device).startswith("cuda") and not torch_sorter and 2 * L >= 64:
        return _build_files_hip(prefix, fwd, str(device), log)
    # ---- .pac ----
    pad = np.zeros((-L) % 4, dtype=np.uint8)
    f4 = np.concatenate([fwd, pad]).reshape(-1, 4)
    pac = ((f4[:, 0] << 6) | (f4[:, 1] << 4) | (f4[:, 2] << 2) | f4[:, 3]).astype(np.uint8)
    with open(prefix + ".pac", "wb") as f:
        f.write(pac.tobytes())
        if L % 4 == 0:
            f.write(b"\0")
        f.write(bytes([L % 4]))
    if log: log("  .pac/.ann/.amb written")
    text = np.concatenate([fwd, (3 - fwd)[::-1]])
    n = 2 * L
    tt = torch.from_numpy(text).to(device)
    if log: log("  text on the device")
    big = torch_sorter == "bucketed" or n >= (3 << 29) or _os.environ.get("DART_SA_BUCKETED") == "1"      # >= 1.6 G symbols: the lean sorter
    sa = suffix_array_bucketed(tt, log) if big else suffix_array(tt)
    if log: log("  suffix array done")
    primary = int(torch.argmin(sa))                           # the row of suffix 0 (nonzero() is limited to < 2^31 elements)
    if log: log("  primary = %d" % primary)
    sa_s = sa[32::32].cpu().numpy().astype(np.uint64)         # rows 32, 64, ... of the (n+1)-row matrix
    if log: log("  SA sampled")
    # BWT with the '$' row removed, gathered in chunks (the index tensor of a one-shot gather is another N int64)
    bwt = torch.empty(n, dtype=torch.uint8, device=tt.device)
    CH = 1 << 28
    for r0 in range(0, n + 1, CH):
        r1 = min(n + 1, r0 + CH)
        prev = tt[(sa[r0:r1] - 1).clamp(min=0)]
        if r0 <= primary < r1:                                # rows after the primary shift up by one
            bwt[r0:primary] = prev[: primary - r0]
            bwt[primary:r1 - 1] = prev[primary - r0 + 1:]
        elif r1 <= primary:
            bwt[r0:r1] = prev
        else:
            bwt[r0 - 1:r1 - 1] = prev
        del prev
    del sa
    if log: log("  BWT gathered")
    cnt = np.array([sum(int((tt[c0:min(n, c0 + (1 << 30))] == c).sum()) for c0 in range(0, n, 1 << 30)) for c in range(4)], dtype=np.uint64)
    del tt
    L2 = np.concatenate([[0], np.cumsum(cnt)]).astype(np.uint64)
    if log: log("  symbol counts done")
    # ---- Occ-interleaved .bwt ----
    nblk = (n + 127) // 128
    blocks, occ_last = _pack_bwt_occ(bwt, n)
    if log: log("  Occ blocks packed")
    del bwt
    flat = blocks.reshape(-1)
    nwords = (n + 15) // 16
    body = flat[: (nblk - 1) * 16 + 8 + (nwords - (nblk - 1) * 8)] if nblk else flat[:0]
    with open(prefix + ".bwt", "wb") as f:
        f.write(np.array([primary], dtype=np.uint64).tobytes())
        f.write(L2[1:5].tobytes())
        f.write(body.tobytes())
        f.write(occ_last.tobytes())
    if log: log("  .bwt written")
    with open(prefix + ".sa", "wb") as f:
        f.write(np.array([primary], dtype=np.uint64).tobytes())
        f.write(L2[1:5].tobytes())
        f.write(np.array([32, n], dtype=np.uint64).tobytes())
        f.write(sa_s[: (n + 32) // 32 - 1].tobytes())
    if log: log("  .sa written")
    return {"l_pac": L, "seq_len": n, "primary": primary}


def build_index_from_fasta(fasta: str, prefix: str, device: str | None = None) -> dict:
    names, annos, seqs = read_fasta(fasta)
    return build_index(prefix, names, annos, seqs, device)


def build_index_from_genome(g, prefix: str, device: str | None = None, log=None) -> dict:
    """g: dart_amd.synth.Genome (no FASTA round trip)."""
    return build_index(prefix, g.names, [""] * len(g.names), [int(x) for x in g.lengths], device, log, codes=g.codes)
