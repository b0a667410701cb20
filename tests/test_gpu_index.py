"""GPU suite (-m gpu): the index builder's GPU path (dart_amd/index_build.py with device="cuda") against the digests of the
REFERENCE indexer's files (tests/golden/manifest.json: made by oracle/_ref/bwt_index, BWT_Index/bwtindex.c:77-148).  The other
GPU tests build fresh genomes with this path and hand the same files to the oracle, so a wrong-but-consistent index would
pass there; here the five files of the three golden genomes must be the reference's bytes, through both suffix sorters."""
import os
import pytest
import common
from dart_amd import synth, index_build

pytestmark = pytest.mark.gpu
CASES = sorted(common.MANIFEST["cases"])


def _select(sorter, monkeypatch):
    """hip = the library's own build (di_build_files: kernels and their driver in libdartindex.so, the default on a GPU and what `dart index` runs);
    hip-python = the same kernels driven from index_build.py; plain / bucketed = the torch-orchestrated sorters kept as cross-checks"""
    monkeypatch.delenv("DART_SA_BUCKETED", raising=False)
    monkeypatch.delenv("DART_INDEX_DRIVER", raising=False)
    if sorter.startswith("hip"):
        monkeypatch.delenv("DART_SA_TORCH", raising=False)
        if sorter == "hip-python":
            monkeypatch.setenv("DART_INDEX_DRIVER", "python")
    else:
        monkeypatch.setenv("DART_SA_TORCH", sorter)


@pytest.mark.parametrize("sorter", ["hip", "hip-python", "plain", "bucketed"])
@pytest.mark.parametrize("name", CASES)
def test_gpu_index_builder_matches_reference_indexer(name, sorter, workdir, monkeypatch):
    import torch
    assert torch.cuda.is_available()
    spec = common.MANIFEST["cases"][name]
    g = synth.make_genome(spec["lengths"], seed=spec["gseed"], repeat_scale=spec["rscale"], n_introns=spec["nintr"])
    _select(sorter, monkeypatch)
    prefix = os.path.join(workdir, "gpuidx_%s_%s" % (name, sorter))
    index_build.build_index_from_genome(g, prefix, device="cuda")
    for ext, want in common.MANIFEST["manifest"][name]["index_sha256"].items():
        assert common.sha(prefix + "." + ext) == want, "GPU-built .%s differs from the reference bwt_index output (%s sorter)" % (ext, sorter)


@pytest.mark.parametrize("sorter", ["hip", "hip-python", "plain", "bucketed"])
@pytest.mark.parametrize("name", sorted(common.MANIFEST["big_index"]))
def test_gpu_index_builder_matches_reference_indexer_at_chr20_size(name, sorter, workdir, monkeypatch):
    """the same at the bench's size class: a 64 444 167 bp chromosome (129 M-symbol text, 2^27 sampled rows ...), the planted-repeat genome of
    `bench.py --genome chr20` and the human-like one (deep repeats: many doubling rounds), each through both suffix sorters -- the GPU-built
    .bwt/.sa/.pac/.ann/.amb must be the bytes the REFERENCE's bwt_index wrote for that genome (digests made by tests/golden/make_golden.py in
    the container that has the reference; the files are ~100 MB).  The GRCh38-sized tests use the same builder on a larger text."""
    import hashlib
    ent = common.MANIFEST["big_index"][name]; spec = ent["spec"]
    g = synth.make_genome(spec["lengths"], seed=spec["gseed"], names=spec["names"], model=spec["model"])
    assert hashlib.sha256(g.codes.tobytes()).hexdigest() == ent["codes_sha256"], "synthetic generator drifted from the genome the reference indexed"
    _select(sorter, monkeypatch)
    prefix = os.path.join(workdir, "gpuidx_%s_%s" % (name, sorter))
    index_build.build_index_from_genome(g, prefix, device="cuda")
    for ext, want in ent["index_sha256"].items():
        assert common.sha(prefix + "." + ext) == want, "GPU-built .%s differs from the reference bwt_index output (%s, %s sorter)" % (ext, name, sorter)
        os.remove(prefix + "." + ext)


def test_gpu_radix_sort_matches_stable_sort():
    """dg_sort_pairs (dart_amd/csrc/dg_sort.h), the index builder's sorter: random keys of several widths and counts (not multiples of
    the 4096-pair tile, heavy duplicates, a single distinct key) against torch's stable sort -- keys AND the order of equal keys"""
    import torch
    g = torch.Generator(device="cuda"); g.manual_seed(7)
    for n, bits, distinct in ((1, 8, None), (2, 1, None), (4095, 13, None), (4097, 38, None), (300001, 63, None), (1000003, 20, 37), (2500000, 38, 1), (777777, 4, None)):
        hi = (1 << bits) if distinct is None else distinct
        key = torch.randint(0, min(hi, (1 << 62)), (n,), dtype=torch.int64, device="cuda", generator=g)
        if bits == 63:
            key = key | (torch.randint(0, 2, (n,), dtype=torch.int64, device="cuda", generator=g) << 62)
        want_k, want_o = torch.sort(key, stable=True)
        got_k, got_o = index_build.sort_pairs(key.clone(), bits)
        assert torch.equal(got_k, want_k), (n, bits)
        assert torch.equal(got_o, want_o), (n, bits, "order of equal keys")


def test_gpu_index_kernels_match_their_contracts():
    """every entry point of include/dartindex.h against tests/index_emul.py's numpy restatement of its contract, on texts whose lengths are
    not multiples of the tile, the word or the block, whose ends meet the '$' inside the key, and on sorted key lists with long runs"""
    import ctypes as C
    import numpy as np
    import torch
    import index_emul
    lib = index_build._index_lib()
    dev = torch.device("cuda:0")
    ops, emu = index_build._HipOps(dev), index_emul.EmulOps()
    rng = np.random.default_rng(5)
    for L in (37, 4096 * 2 + 1, 100003, 262144):
        fwd = rng.integers(0, 4, L).astype(np.uint8)
        fwd[:35] = 3                                              # the text ends in a run of A
        n = 2 * L
        T_want = index_emul.pack_text(fwd)
        T = torch.empty(int(lib.di_text_words(n)), dtype=torch.int64, device=dev)
        f = torch.from_numpy(fwd).to(dev)
        index_build._di(lib.di_pack_text(0, f.data_ptr(), L, T.data_ptr()), "di_pack_text")
        assert torch.equal(T.cpu(), T_want), ("di_pack_text", L)
        tiles = (n + 1 + 4095) // 4096
        tab = torch.empty(16 * tiles, dtype=torch.int32, device=dev)
        tab_want = torch.empty(16 * tiles, dtype=torch.int32)
        ops.bucket_hist(T, n, tab); emu.bucket_hist(T_want, n, tab_want)
        assert torch.equal(tab.cpu(), tab_want), ("di_bucket_hist", L)
        for pair in range(16):
            r = tab_want.view(16, tiles)[pair]
            m = int(r.sum())
            base = (torch.cumsum(r, 0) - r).to(torch.int32)
            k, v = torch.zeros(m + 1, dtype=torch.int64, device=dev), torch.zeros(m + 1, dtype=torch.int64, device=dev)
            kw, vw = torch.zeros(m + 1, dtype=torch.int64), torch.zeros(m + 1, dtype=torch.int64)
            ops.bucket_keys(T, n, pair, base.to(dev), k, v); emu.bucket_keys(T_want, n, pair, base, kw, vw)
            assert torch.equal(k.cpu(), kw) and torch.equal(v.cpu(), vw), ("di_bucket_keys", L, pair)
        # the BWT words and per-block counts of a made-up row order
        N = n + 1
        sa_h = rng.permutation(N).astype(np.int64)
        primary = int(np.nonzero(sa_h == 0)[0][0])
        sa = torch.from_numpy(sa_h).to(dev)
        nblk = (n + 127) // 128
        blocks = torch.zeros(nblk * 16, dtype=torch.int32, device=dev)
        c4 = torch.zeros(nblk, dtype=torch.int32, device=dev)
        index_build._di(lib.di_bwt_blocks(0, sa.data_ptr(), T.data_ptr(), n, primary, blocks.data_ptr(), c4.data_ptr()), "di_bwt_blocks")
        text = np.concatenate([fwd, (3 - fwd)[::-1]]).astype(np.int64)
        rows = np.delete(sa_h, primary)
        bwt = np.zeros(nblk * 128, dtype=np.int64)
        bwt[:n] = text[rows - 1]
        words = (bwt.reshape(-1, 16) << (30 - 2 * np.arange(16))).sum(axis=1).astype(np.uint32)
        got = blocks.cpu().numpy().view(np.uint32).reshape(nblk, 16)
        assert np.array_equal(got[:, 8:].reshape(-1), words), ("di_bwt_blocks words", L)
        valid = (np.arange(nblk * 128) < n).reshape(nblk, 128)
        per = np.stack([((bwt.reshape(nblk, 128) == c) & valid).sum(axis=1) for c in range(4)], axis=1)
        c4h = c4.cpu().numpy().view(np.uint32)
        assert np.array_equal(np.stack([(c4h >> (8 * c)) & 255 for c in range(4)], axis=1), per), ("di_bwt_blocks counts", L)
    # regroup and the rank-pair keys, on key lists with runs that cross lanes, waves and tiles
    for m, distinct in ((1, 1), (17, 3), (4096, 50), (4097, 4097 * 4), (150001, 9000), (150001, 1), (70000, 1 << 40)):
        kh = np.sort(rng.integers(0, distinct, m).astype(np.uint64))
        for with_pos in (False, True):
            span = m * 3 if with_pos else m
            lo = 1234567
            total = lo + span + 10
            vh = rng.permutation(total)[:m].astype(np.int64)
            ph = np.sort(rng.permutation(span)[:m]).astype(np.int32) if with_pos else None
            out = []
            for o, d in ((ops, dev), (emu, torch.device("cpu"))):
                keys, vals = torch.from_numpy(kh.view(np.int64)).to(d), torch.from_numpy(vh).to(d)
                pos = torch.from_numpy(ph).to(d) if with_pos else None
                rank = torch.full((total,), -1, dtype=torch.int64, device=d)
                sa = torch.full((total,), -1, dtype=torch.int64, device=d)
                new_pos = torch.full((m,), -1, dtype=torch.int32, device=d)
                scratch = torch.empty(2 * ((m + 4095) // 4096) + 4, dtype=torch.int32, device=d)
                t = o.regroup(keys, vals, pos, m, lo, rank, sa, new_pos, scratch)
                out.append((t, rank.cpu(), sa.cpu(), new_pos[:t].cpu()))
            assert out[0][0] == out[1][0], ("di_regroup tied count", m, distinct, with_pos)
            for a, b, what in zip(out[0][1:], out[1][1:], ("rank", "sa", "new_pos")):
                assert torch.equal(a, b), ("di_regroup", what, m, distinct, with_pos)
    N = 500000
    sa_h = rng.permutation(N).astype(np.int64)
    rank_h = rng.integers(0, N, N).astype(np.int64)
    lo, m = 1000, 30011
    rank_h[sa_h[lo:lo + 3 * m]] = lo + rng.integers(0, 3 * m, 3 * m)          # members of the bucket rank inside it
    ph = np.sort(rng.permutation(3 * m)[:m]).astype(np.int32)
    res = []
    for o, d in ((ops, dev), (emu, torch.device("cpu"))):
        keys, vals = torch.zeros(m, dtype=torch.int64, device=d), torch.zeros(m, dtype=torch.int64, device=d)
        o.doubling_keys(torch.from_numpy(sa_h).to(d), torch.from_numpy(rank_h).to(d), lo, torch.from_numpy(ph).to(d), m, 400000, N, N.bit_length(), keys, vals)
        res.append((keys.cpu(), vals.cpu()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]), "di_doubling_keys"
