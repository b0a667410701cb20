"""GPU suite (-m gpu): the HIP path through the C ABI against the oracle and the golden vectors."""
import gzip, os
import numpy as np
import pytest
import common, oracle_py
from dart_amd import host, synth, index_build

pytestmark = pytest.mark.gpu
CASES = sorted(common.MANIFEST["cases"])


@pytest.fixture(scope="module")
def ctxs(workdir):
    out = {}
    for name in CASES:
        c = common.build_case(name, workdir)
        ix = host.Index(c["prefix"])
        out[name] = (c, ix, host.DartGPU(ix), oracle_py.Oracle(c["prefix"]))
    yield out
    for c, ix, gpu, orc in out.values():
        gpu.close(); orc.close()


cigars_of, assert_same = common.cigars_of, common.assert_same


@pytest.mark.parametrize("name", CASES)
def test_gpu_matches_golden_sam(name, ctxs):
    c, ix, gpu, orc = ctxs[name]
    so, rl, flat = host.pack_reads(c["reads"])
    for run in c["runs"]:
        p, h = common.parse_flags(run["flags"])
        gpu.set_params(host.default_params(paired=int(c["spec"]["paired"]), **p))
        res = gpu.map_batch(so, rl, flat)
        text, junc = common.records_to_text(c, p, h, res.reads, res.reports, res.cigar, res.sj, ix)
        want = common.golden_sam(run["base"])
        assert text == want, common.first_diff(text, want)
        assert junc == common.golden_junctions(run["base"])
        assert_same(res, orc.map_batch(orc.params(paired=int(c["spec"]["paired"]), **p), so, rl, flat))


def _nw_vectors():
    out = []
    for fn in ("nw_known_answers.tsv.gz", "nw_known_answers_large.tsv.gz"):
        for line in gzip.open(os.path.join(common.GOLDEN, fn), "rt"):
            a, b, o1, o2 = line.rstrip("\n").split("\t")
            out.append(((a.encode(), b.encode()), (o1.encode(), o2.encode())))
    return out


def test_gpu_nw_known_answers(ctxs):
    """every form of nw_alignment that runs in production against the reference-generated known answers (nw_alignment.cpp:18-82):
    0 = serial strips, 1 = register strips (<= 24 x 24), 2 = the wave-wide service (8-lane groups / whole wave) + the owner's
    traceback, 3 = the whole-wave form for every pair.  Forms 2 and 3 read the genome side from 2-bit text: ACGT vectors only."""
    c, ix, gpu, orc = ctxs["se100"]
    vec = _nw_vectors()
    acgt = set(b"ACGT")
    sel = {0: vec,
           1: [v for v in vec if 1 <= len(v[0][0]) <= 24 and 1 <= len(v[0][1]) <= 24],
           2: [v for v in vec if set(v[0][1]) <= acgt],
           3: [v for v in vec if set(v[0][1]) <= acgt]}
    assert len(sel[0]) == 1508 + 420 and len(sel[1]) > 300 and len(sel[2]) > 1200
    assert sum(len(v[0][1]) > 64 for v in sel[2]) > 250 and sum(len(v[0][1]) > 128 for v in sel[2]) > 100     # several 64-column blocks
    for mode, vs in sel.items():
        got = gpu.probe_nw([v[0] for v in vs], mode=mode)
        bad = [i for i, (g, v) in enumerate(zip(got, vs)) if g != v[1]]
        assert not bad, "nw form %d: %d of %d pairs differ, first %r -> %r, expected %r" % (mode, len(bad), len(vs), vs[bad[0]][0], got[bad[0]], vs[bad[0]][1])
    with pytest.raises(RuntimeError):
        gpu.probe_nw([(b"ACGT", b"ACNT")], mode=2)


def test_gpu_seeds_match_oracle(ctxs):
    for name in CASES:
        c, ix, gpu, orc = ctxs[name]
        so, rl, flat = host.pack_reads(c["reads"])
        gpu.set_params(host.default_params())
        gso, grp, gsl, ggp = gpu.probe_seeds(so, rl, flat)
        for i in range(0, len(rl), 7):
            rp, sl, gp = orc.seeds(orc.params(), c["reads"][i].tobytes())
            a, b = gso[i], gso[i + 1]
            assert np.array_equal(rp, grp[a:b]) and np.array_equal(sl, gsl[a:b]) and np.array_equal(gp, ggp[a:b]), (name, i)


def test_gpu_seed_queue_configurations(ctxs, monkeypatch):
    """the queue kernels of dg_seedq.h -- k_seed_qf (free-running waves, the default) and k_seed_q (barrier phases, DG_SEED_PHASES=1) --
    against the lane-per-read kernel k_seed: the same hits whatever the number of read slots per workgroup (64 ... 512: the queue rings
    wrap thousands of times, most chunks are partial), the waves per workgroup, the workgroups per CU, the size below which a wave
    would rather wait than take a partial chunk, and the bail-out threshold that moves reads to k_seed_heavy; ragged reads, N runs,
    reads shorter than a seed; then whole records against the oracle"""
    c, ix, gpu, orc = ctxs["pe101_spliced"]
    rng = np.random.default_rng(11)
    asc = c["genome"].ascii()
    seqs = []
    for i in range(5000):
        L = int(rng.choice([8, 15, 16, 31, 64, 101, 101, 101, 150, 333, 480]))
        p = int(rng.integers(0, c["genome"].total - 600))
        s = bytearray(asc[p:p + L].tobytes())
        for k in rng.integers(0, L, size=L // 50): s[int(k)] = int(rng.choice(list(b"ACGT")))
        if i % 9 == 0 and L > 40: s[L // 3:L // 3 + 5] = b"NNNNN"
        if i % 13 == 0: s = bytearray(bytes(rng.choice(list(b"ACGT"), L).astype(np.uint8)))            # maps nowhere: every start fails
        seqs.append(bytes(s))
    so, rl, flat = host.pack_reads(seqs)
    monkeypatch.setenv("DG_SEED_LEGACY", "1")
    gpu.set_params(host.default_params())                    # (the switches are read by dg_set_params)
    want = gpu.probe_seeds(so, rl, flat)
    monkeypatch.setenv("DG_SEED_LEGACY", "0")
    for phases in ("0", "1"):
        monkeypatch.setenv("DG_SEED_PHASES", phases)
        for lg, wgs, bail, nw, part, multi in ((9, 0, 128, 4, 32, 4), (6, 1, 128, 4, 1, 2), (7, 3, 128, 2, 64, 3), (8, 2, 6, 8, 16, 4), (9, 2, 1000, 1, 32, 1), (10, 1, 128, 8, 48, 0)):
            if phases == "1" and lg > 9:
                continue
            monkeypatch.setenv("DG_SEED_SLOTS_LG", str(lg)); monkeypatch.setenv("DG_SEED_WGS", str(wgs)); monkeypatch.setenv("DG_SEED_BAIL_TRIPS", str(bail))
            monkeypatch.setenv("DG_SEED_WG_WAVES", str(nw)); monkeypatch.setenv("DG_SEED_PARTIAL_MIN", str(part))
            monkeypatch.setenv("DG_SEED_MULTI", str(multi))          # rows of an interval k_seed_qf locates and compares at once (0: Occ steps down to one row)
            gpu.set_params(host.default_params())
            got = gpu.probe_seeds(so, rl, flat)
            for a, b in zip(want, got):
                assert np.array_equal(a, b), (phases, lg, wgs, bail, nw, part, multi)
        monkeypatch.setenv("DG_SEED_SLOTS_LG", "6"); monkeypatch.setenv("DG_SEED_WGS", "2"); monkeypatch.delenv("DG_SEED_BAIL_TRIPS")
        gpu.set_params(host.default_params(paired=0, max_mismatch=4))
        assert_same(gpu.map_batch(so, rl, flat), orc.map_batch(orc.params(paired=0, max_mismatch=4), so, rl, flat))
    for k in ("DG_SEED_SLOTS_LG", "DG_SEED_WGS", "DG_SEED_LEGACY", "DG_SEED_PHASES", "DG_SEED_WG_WAVES", "DG_SEED_PARTIAL_MIN", "DG_SEED_MULTI"):
        monkeypatch.delenv(k)
    gpu.set_params(host.default_params())


def test_gpu_literal_dashes_in_read_gaps_and_segment_pairs(ctxs):
    """A literal '-' in a read changes what the reference's string code makes of an alignment column (AddNewCigarElements, tools.cpp:49-104, and FillGapsBetweenAdjacentSeeds
    look at the characters), so the forms that never build the strings -- d_gap_small, d_gap_right_tb / d_gap_left_tb, d_process_pair_tb -- hand such gaps and pairs to the string
    forms (d_has_dash).  Spliced and indel reads of 101 and 250 bases with dashes, lower case and N planted at random places, next to the junctions, and inside stretches of
    noise between two seeds (wide gaps, large pairs): every record against the oracle."""
    c, ix, gpu, orc = ctxs["pe101_spliced"]
    rng = np.random.default_rng(77)
    seqs = []
    for rlen, n_pairs, seed in ((101, 6000, 501), (250, 2500, 502)):
        m1, m2 = synth.make_reads(c["genome"], n_pairs, rlen=rlen, seed=seed, spliced_frac=0.6, indel_frac=0.3, n_frac=0.0)
        for i in range(n_pairs):
            for m in (m1, m2):
                s = bytearray(m[i].tobytes())
                k = i % 8
                if k < 3:                                        # one to three characters anywhere: some land in the few bases between two exons' seeds
                    for q in rng.integers(0, rlen, size=k + 1): s[int(q)] = ord("-")
                elif k == 3:                                     # a stretch of noise with a dash in it: a wide gap or a large pair, depending on what lies around it
                    a = int(rng.integers(20, rlen - 70)); w = int(rng.integers(26, 60))
                    s[a:a + w] = bytes(rng.choice(list(b"ACGT"), w).astype(np.uint8)); s[a + w // 2] = ord("-")
                elif k == 4:                                     # the same without the dash: the forms on the bits
                    a = int(rng.integers(20, rlen - 70)); w = int(rng.integers(26, 60))
                    s[a:a + w] = bytes(rng.choice(list(b"ACGT"), w).astype(np.uint8))
                elif k == 5:                                     # lower case and N instead: characters that differ from the genome's without being a gap
                    for q in rng.integers(0, rlen, size=4): s[int(q)] = s[int(q)] | 0x20
                    s[int(rng.integers(0, rlen))] = ord("N")
                seqs.append(bytes(s))
    so, rl, flat = host.pack_reads(seqs)
    for mis in (5, 12):
        gpu.set_params(host.default_params(paired=1, max_mismatch=mis))
        assert_same(gpu.map_batch(so, rl, flat), orc.map_batch(orc.params(paired=1, max_mismatch=mis), so, rl, flat))
    ctr = gpu.counters()
    assert ctr["nw_calls"] > 1000
    gpu.set_params(host.default_params())


def test_gpu_edge_cases(ctxs):
    """empty batch, ragged lengths, reads shorter than a seed, all-N reads, lower case, odd paired batch"""
    c, ix, gpu, orc = ctxs["pe101_spliced"]
    gpu.set_params(host.default_params(paired=1, max_mismatch=5))
    res = gpu.map_batch(np.zeros(0, np.uint32), np.zeros(0, np.uint16), np.zeros(0, np.uint8))
    assert len(res.reads) == 0 and len(res.reports) == 0
    rng = np.random.default_rng(3)
    asc = c["genome"].ascii()
    seqs = []
    for i in range(600):
        L = int(rng.choice([1, 5, 13, 14, 16, 17, 30, 50, 75, 101, 130, 200, 250]))
        p = int(rng.integers(0, c["genome"].total - 300))
        s = bytearray(asc[p:p + L].tobytes())
        k = i % 6
        if k == 1: s = bytearray(b"N" * L)
        elif k == 2: s = bytearray(bytes(s).lower())
        elif k == 3 and L > 20: s[L // 2] = ord("N"); s[3] = ord("n")
        elif k == 4 and L > 40: s = s[:20] + bytearray(b"ACGTTGCA") + s[20:]
        seqs.append(bytes(s))
    so, rl, flat = host.pack_reads(seqs)
    for paired in (0, 1):
        gpu.set_params(host.default_params(paired=paired, max_mismatch=3))
        assert_same(gpu.map_batch(so, rl, flat), orc.map_batch(orc.params(paired=paired, max_mismatch=3), so, rl, flat))
    # long reads whose seeds are > 263 read bases apart: re-seeding takes the serial in-kernel path
    long_reads = []
    for i in range(120):
        p = int(rng.integers(1000, c["genome"].total - 120000))
        D = int(rng.choice([700, 5000, 90000]))
        gap = int(rng.integers(270, 420))
        A = asc[p:p + 60].tobytes(); B = asc[p + 60 + gap + D:p + 120 + gap + D].tobytes()
        if i % 3 == 0: X = bytes(rng.choice(list(b"ACGT"), gap).astype(np.uint8))                        # nothing to find
        elif i % 3 == 1: X = asc[p + 60 + D:p + 60 + D + gap].tobytes()                                    # the gap continues after the jump
        else: X = asc[p + 60:p + 60 + gap // 2].tobytes() + bytes(rng.choice(list(b"ACGT"), gap - gap // 2).astype(np.uint8))
        long_reads.append(A + X + B)
    so2, rl2, flat2 = host.pack_reads(long_reads)
    gpu.set_params(host.default_params(paired=0, max_mismatch=8))
    assert_same(gpu.map_batch(so2, rl2, flat2), orc.map_batch(orc.params(paired=0, max_mismatch=8), so2, rl2, flat2))
    assert gpu.counters()["reseed_calls"] > 0
    # reads up to DG_MAX_RLEN: k_seed without LDS staging, small persistent grid for the R^2 workspace
    very_long = []
    for i in range(40):
        L = int(rng.choice([600, 800, 1000]))
        p = int(rng.integers(1000, c["genome"].total - 3000))
        s = bytearray(asc[p:p + L].tobytes())
        for k in rng.integers(0, L, size=L // 40): s[int(k)] = int(rng.choice(list(b"ACGT")))
        if i % 4 == 0: s = s[:L // 2] + bytearray(asc[p + L // 2 + 3000:p + L + 3000].tobytes())          # a 3 kb deletion in the middle
        very_long.append(bytes(s))
    so3, rl3, flat3 = host.pack_reads(very_long)
    gpu.set_params(host.default_params(paired=0, max_mismatch=40))
    assert_same(gpu.map_batch(so3, rl3, flat3), orc.map_batch(orc.params(paired=0, max_mismatch=40), so3, rl3, flat3))
    too_long = [bytes(asc[100:1101].tobytes())]
    with pytest.raises(RuntimeError):
        gpu.map_batch(*host.pack_reads(too_long))
    # odd count in paired mode is mapped read by read (Mapping.cpp:598)
    gpu.set_params(host.default_params(paired=1, max_mismatch=3))
    assert_same(gpu.map_batch(so[:-1], rl[:-1], flat), orc.map_batch(orc.params(paired=1, max_mismatch=3), so[:-1], rl[:-1], flat))


def test_gpu_medium_batch_all_flag_sets(workdir):
    """a fresh 3 Mbp genome, 20 k pairs, every flag set of SURVEY 8c"""
    g = synth.make_genome([2000000, 1000000], seed=31, repeat_scale=50.0, n_introns=400)
    prefix = os.path.join(workdir, "medium")
    index_build.build_index_from_genome(g, prefix)
    ix = host.Index(prefix); gpu = host.DartGPU(ix); orc = oracle_py.Oracle(prefix)
    m1, m2 = synth.make_reads(g, 20000, rlen=101, seed=32, spliced_frac=0.2, indel_frac=0.05, n_frac=0.01)
    so, rl, flat = host.pack_reads(host.interleave_pairs(m1, m2))
    for flags in ([], ["-mis", "5"], ["-mis", "5", "-m"], ["-mis", "2", "-all_sj", "-max_dup", "1000"], ["-mis", "5", "-min_intron", "10", "-max_intron", "200000"]):
        p, _ = common.parse_flags(flags)
        gpu.set_params(host.default_params(paired=1, **p))
        assert_same(gpu.map_batch(so, rl, flat), orc.map_batch(orc.params(paired=1, **p), so, rl, flat, threads=16))
    gpu.close(); orc.close()


def test_gpu_size_independent_properties(workdir):
    """full-size style properties: batch-split invariance and pair-order invariance"""
    g = synth.make_genome([1500000], seed=41, repeat_scale=30.0, n_introns=100)
    prefix = os.path.join(workdir, "props")
    index_build.build_index_from_genome(g, prefix)
    ix = host.Index(prefix); gpu = host.DartGPU(ix, host.default_params(paired=1, max_mismatch=5))
    m1, m2 = synth.make_reads(g, 30000, rlen=101, seed=42, spliced_frac=0.1)
    arr = host.interleave_pairs(m1, m2)
    so, rl, flat = host.pack_reads(arr)
    whole = gpu.map_batch(so, rl, flat)
    # mapping two halves separately gives the same per-read records
    h = 30000
    a = gpu.map_batch(*host.pack_reads(arr[:h])); b = gpu.map_batch(*host.pack_reads(arr[h:]))
    for f in ("score", "sub_score", "mis_num", "mapq", "n_rep", "best", "n_sj"):
        assert np.array_equal(np.concatenate([a.reads[f], b.reads[f]]), whole.reads[f])
    assert np.array_equal(np.concatenate([cigars_of(a.reports, a.cigar), cigars_of(b.reports, b.cigar)]), cigars_of(whole.reports, whole.cigar))
    # permuting the pairs permutes the records
    perm = np.random.default_rng(1).permutation(30000)
    idx = np.stack([2 * perm, 2 * perm + 1], 1).reshape(-1)
    pres = gpu.map_batch(*host.pack_reads(arr[idx]))
    for f in ("score", "sub_score", "mis_num", "mapq", "n_rep", "best"):
        assert np.array_equal(pres.reads[f], whole.reads[f][idx])
    gpu.close()


def test_gpu_records_as_torch_tensor_for_rccl(ctxs):
    """bench.py hands the per-read records to torch.distributed straight from HBM (dg_batch_device_ptrs)"""
    import torch
    c, ix, gpu, orc = ctxs["se100"]
    so, rl, flat = host.pack_reads(c["reads"])
    gpu.set_params(host.default_params(paired=0, max_mismatch=3))
    res = gpu.map_batch(so, rl, flat)
    t = gpu.device_reads_tensor()
    assert t.is_cuda and tuple(t.shape) == (len(rl), host.READ_OUT.itemsize)
    back = t.cpu().numpy().reshape(-1).view(host.READ_OUT)
    assert np.array_equal(back, res.reads)


def test_gpu_cloned_contexts_run_concurrently(workdir, monkeypatch):
    """dg_clone: contexts sharing one index, one host thread each, different batches in flight at once (bench.py's
    pipeline); every context's records equal the oracle's.  The contexts' re-seeding kernels (a sixth of these reads span an intron) run on streams the
    contexts SHARE (DG_S2_SHARED, 3 by default; dg_api.hip make_ctx_objects): five contexts on three, on one, and on private streams (0)."""
    import threading
    g = synth.make_genome([1200000, 800000], seed=51, repeat_scale=40.0, n_introns=200)
    prefix = os.path.join(workdir, "clones")
    index_build.build_index_from_genome(g, prefix)
    ix = host.Index(prefix); orc = oracle_py.Oracle(prefix)
    n_ctx = 5
    batches, want = [], []
    for j in range(n_ctx):
        m1, m2 = synth.make_reads(g, 9000 + 2000 * j, rlen=101, seed=52 + j, spliced_frac=0.15, indel_frac=0.04, n_frac=0.01)
        batches.append(host.pack_reads(host.interleave_pairs(m1, m2)))
        want.append(orc.map_batch(orc.params(paired=1, max_mismatch=5), *batches[j], threads=16))
    for shared in (None, "1", "0"):
        if shared is None: monkeypatch.delenv("DG_S2_SHARED", raising=False)
        else: monkeypatch.setenv("DG_S2_SHARED", shared)
        gpu = host.DartGPU(ix, host.default_params(paired=1, max_mismatch=5))
        ctx = [gpu] + [gpu.clone() for _ in range(n_ctx - 1)]
        out, errs = [None] * n_ctx, []
        def work(j):
            try:
                for _ in range(3):                       # several rounds so that the batches really overlap
                    out[j] = ctx[j].map_batch(*batches[j])
            except Exception as e:
                errs.append(e)
        th = [threading.Thread(target=work, args=(j,)) for j in range(n_ctx)]
        for t in th: t.start()
        for t in th: t.join()
        assert not errs, (shared, errs)
        for j in range(n_ctx):
            assert_same(out[j], want[j])
        gpu.close()
    monkeypatch.delenv("DG_S2_SHARED", raising=False)
    orc.close()


def test_gpu_reference_equivalent_counters(workdir):
    """The algorithmic-byte accounting: the kernels skip most of the reference's memory accesses (prefix
    table, full SA, direct text comparison) but memoise what the reference would have done -- those counters must equal
    the oracle's, which counts while really executing the reference algorithm."""
    g = synth.make_genome([3000000], seed=5, repeat_scale=20.0)
    prefix = os.path.join(workdir, "ctr")
    index_build.build_index_from_genome(g, prefix)
    ix = host.Index(prefix); orc = oracle_py.Oracle(prefix)
    m1, m2 = synth.make_reads(g, 20000, rlen=101, seed=6)
    so, rl, flat = host.pack_reads(host.interleave_pairs(m1, m2))
    orc.map_batch(orc.params(paired=1, max_mismatch=5), so, rl, flat, threads=8)
    oc = orc.counters
    gpu = host.DartGPU(ix, host.default_params(paired=1, max_mismatch=5))
    gpu.map_batch(so, rl, flat)
    c = gpu.counters()
    assert c["steps"] == oc["n_2occ4"]
    assert c["lf_steps"] == oc["n_lf"]
    assert c["sa_lookups"] == oc["n_sa"]
    assert c["nw_calls"] == oc["n_nw"] and c["nw_cells"] == oc["nw_cells"]
    assert c["reseed_calls"] == oc["n_reseed"] and c["reseed_window"] == oc["reseed_window"]
    # Occ blocks: inside a direct text comparison every base is charged one block; the reference fetches two when the
    # interval's two rows straddle a 128-row block (1 in 128 steps), so the figure is a lower bound, < 1 % under
    assert 0 <= oc["n_occ_blocks"] - (c["occ_blocks"] + c["lf_steps"]) <= 1e-2 * oc["n_occ_blocks"]
    gpu.close(); orc.close()


def test_gpu_full_size_batch_matches_oracle(workdir):
    """BASELINE configs[1] at full size: chr20-sized genome, 1 M pairs 2x101 in ONE batch, every record field, CIGAR op and
    splice-junction tuple against the oracle (16 threads, a few seconds)."""
    import bench
    cache = os.path.join(workdir, "bench_cache")
    prefix, g = bench.prepare_index(cache, bench.CHR20_LEN, 0, lambda: None)
    m1, m2 = synth.make_reads(g, 1000000, rlen=101, seed=1000, sub_rate=0.01, indel_frac=0.02, n_frac=0.002)
    so, rl, flat = host.pack_reads(host.interleave_pairs(m1, m2))
    ix = host.Index(prefix); orc = oracle_py.Oracle(prefix)
    gpu = host.DartGPU(ix, host.default_params(paired=1, max_mismatch=5))
    assert_same(gpu.map_batch(so, rl, flat), orc.map_batch(orc.params(paired=1, max_mismatch=5), so, rl, flat, threads=16))
    gpu.close(); orc.close()


def test_gpu_spliced_2x151_batch_matches_oracle(workdir):
    """BASELINE configs[4] shape on the chr20-sized genome: 2x151, 30 % of the reads span a planted intron of 200 b - 500 kb
    (half with GT..AG), -max_intron 500000: the long-gap re-seeding and NW stress, 300 k pairs against the oracle."""
    import bench
    cache = os.path.join(workdir, "bench_cache")
    prefix, g = bench.prepare_index(cache, bench.CHR20_LEN, 0, lambda: None, 20000)
    m1, m2 = synth.make_reads(g, 300000, rlen=151, seed=1001, sub_rate=0.01, indel_frac=0.02, n_frac=0.002, spliced_frac=0.3)
    so, rl, flat = host.pack_reads(host.interleave_pairs(m1, m2))
    ix = host.Index(prefix); orc = oracle_py.Oracle(prefix)
    gpu = host.DartGPU(ix, host.default_params(paired=1, max_mismatch=5, max_intron=500000))
    res = gpu.map_batch(so, rl, flat)
    assert_same(res, orc.map_batch(orc.params(paired=1, max_mismatch=5, max_intron=500000), so, rl, flat, threads=16))
    assert len(res.sj) > 10000                       # the junction path really ran
    gpu.close(); orc.close()


def test_gpu_reseed_windows_shared_by_several_waves(workdir, monkeypatch):
    """k_reseed splits a genome window of more than 32 768 diagonals into chunks that different waves scan; the wave that finishes the
    last chunk replays the chunks' reported diagonals through the reference's scan (KmerAnalysis.cpp:146-163).  Introns of up to 500 kb
    on a 6 Mbp genome, 2x151 and 2x101.  The default chunk size; 4096-diagonal chunks (the test hook: up to 120 chunks per window);
    chunk records that hold 0 or 1 entries inline (every chunk with a hit then sends its entries through the pool of 64-entry blocks);
    a pool of two blocks (exhausted at once: such windows are scanned again whole by their last wave) -- all must give the oracle's
    records and its re-seeding counters."""
    g = synth.make_genome([4000000, 2000000], seed=61, repeat_scale=30.0, n_introns=1500)
    prefix = os.path.join(workdir, "rs_shared")
    index_build.build_index_from_genome(g, prefix)
    ix = host.Index(prefix); orc = oracle_py.Oracle(prefix)
    gpu = host.DartGPU(ix)
    for rlen, n_pairs in ((151, 40000), (101, 40000)):
        m1, m2 = synth.make_reads(g, n_pairs, rlen=rlen, seed=62 + rlen, sub_rate=0.01, indel_frac=0.02, n_frac=0.002, spliced_frac=0.5)
        so, rl, flat = host.pack_reads(host.interleave_pairs(m1, m2))
        want = orc.map_batch(orc.params(paired=1, max_mismatch=5, max_intron=500000), so, rl, flat, threads=16)
        oc = dict(orc.counters)
        seen = {}
        for chunk, inline, pool in ((None, None, None), (4096, None, None), (4096, 0, None), (8192, 1, None), (4096, 0, 2), (65536, None, None)):
            for k, v in (("DG_RS_CHUNK", chunk), ("DG_RS_ENT_MAX", inline), ("DG_RS_POOL_BLOCKS", pool)):
                if v is None: monkeypatch.delenv(k, raising=False)
                else: monkeypatch.setenv(k, str(v))
            gpu.set_params(host.default_params(paired=1, max_mismatch=5, max_intron=500000))      # (reads the DG_* switches)
            assert_same(gpu.map_batch(so, rl, flat), want)
            c = gpu.counters()
            assert c["reseed_calls"] == oc["n_reseed"] and c["reseed_window"] == oc["reseed_window"], (chunk, inline, pool)
            seen[(chunk, inline, pool)] = (c["k_reseed_items"], c["k_reseed_windows_scanned_again_whole"], c["k_reseed_chunks_through_pool"], c["reseed_calls"])
        dflt, small, pooled, starved = seen[(None, None, None)], seen[(4096, None, None)], seen[(4096, 0, None)], seen[(4096, 0, 2)]
        assert dflt[0] > dflt[3] > 300, seen              # windows longer than a chunk occurred ...
        assert small[0] > 3 * dflt[3], seen               # ... and the small chunks multiplied the items
        assert dflt[1] == 0 and small[1] == 0 and pooled[2] > 100 and pooled[2] > small[2] and pooled[1] < starved[1], seen      # the pool carried what the records could not hold
        assert starved[1] > 50, seen                      # and without a pool the last wave did the window again
    gpu.close(); orc.close()


def test_gpu_second_stream_for_the_reseeding_kernels_still_gives_the_same_records(workdir, monkeypatch):
    """DG_ONE_STREAM=0 (rounds 2-4's arrangement, kept as a switch): k_reseed on a second stream beside the report of the candidates without
    re-seeding jobs, the others in a second k_report launch behind it.  Same records as the oracle's, and as the default's (one stream, k_reseed in
    front of one k_report launch), on a batch where a third of the units have jobs."""
    g = synth.make_genome([3000000, 1500000], seed=71, repeat_scale=30.0, n_introns=800)
    prefix = os.path.join(workdir, "two_streams")
    index_build.build_index_from_genome(g, prefix)
    ix = host.Index(prefix); orc = oracle_py.Oracle(prefix)
    m1, m2 = synth.make_reads(g, 30000, rlen=125, seed=72, sub_rate=0.01, indel_frac=0.03, n_frac=0.002, spliced_frac=0.4)
    so, rl, flat = host.pack_reads(host.interleave_pairs(m1, m2))
    want = orc.map_batch(orc.params(paired=1, max_mismatch=5, max_intron=200000), so, rl, flat, threads=16)
    for one in ("0", "1"):
        monkeypatch.setenv("DG_ONE_STREAM", one)
        gpu = host.DartGPU(ix, host.default_params(paired=1, max_mismatch=5, max_intron=200000))       # (the switch is read when a context is created)
        res = gpu.map_batch(so, rl, flat)
        assert_same(res, want)
        names = [n for n, _ in gpu.timings()]
        assert ("k_report_jobs" in names) == (one == "0"), names
        assert gpu.counters()["reseed_calls"] == orc.counters["n_reseed"] > 100
        gpu.close()
    orc.close()


def test_gpu_noisy_long_reads_wave_nw_paths(workdir):
    """Stress of the wave-wide alignment service and of the wave-per-read layout: a repeat-rich 3 Mbp genome, 2x250 reads with
    4 % substitutions and an indel in a third of them (segment pairs wider than 64 columns: one pair per wave; up to 64: eight
    per wave; gap filling on spliced reads), reads from repeat families (dozens of candidates), then 76-base single-end
    reads; every record against the oracle."""
    g = synth.make_genome([2000000, 1000000], seed=51, repeat_scale=200.0, n_introns=600)
    prefix = os.path.join(workdir, "noisy")
    index_build.build_index_from_genome(g, prefix)
    ix = host.Index(prefix); gpu = host.DartGPU(ix); orc = oracle_py.Oracle(prefix)
    m1, m2 = synth.make_reads(g, 30000, rlen=250, seed=52, sub_rate=0.04, indel_frac=0.35, spliced_frac=0.25, n_frac=0.01, frag_mean=600.0)
    so, rl, flat = host.pack_reads(host.interleave_pairs(m1, m2))
    for flags in (["-mis", "30"], ["-mis", "30", "-m", "-max_dup", "1000"]):
        p, _ = common.parse_flags(flags)
        gpu.set_params(host.default_params(paired=1, **p))
        res = gpu.map_batch(so, rl, flat)
        assert_same(res, orc.map_batch(orc.params(paired=1, **p), so, rl, flat, threads=16))
        assert gpu.counters()["nw_cells"] > 50 * gpu.counters()["nw_calls"]        # large matrices really occurred
    s1, _ = synth.make_reads(g, 40000, rlen=76, seed=53, sub_rate=0.03, indel_frac=0.2, spliced_frac=0.2, paired=False)
    so, rl, flat = host.pack_reads(s1)
    gpu.set_params(host.default_params(paired=0, max_mismatch=10))
    assert_same(gpu.map_batch(so, rl, flat), orc.map_batch(orc.params(paired=0, max_mismatch=10), so, rl, flat, threads=16))
    gpu.close(); orc.close()


def test_gpu_both_seeding_kernels_match_oracle(workdir, monkeypatch):
    """the free-running queue kernel k_seed_qf (default), the phased one k_seed_q (DG_SEED_PHASES=1) and the lane-per-read kernel k_seed
    (DG_SEED_LEGACY=1; also what reads longer than 496 bases take): same records as the oracle and the same reference-equivalent
    counters, on plain, spliced and N-rich reads of a repeat-rich genome (many reads end in k_seed_heavy)."""
    g = synth.make_genome([2000000, 1000000], seed=61, repeat_scale=100.0, n_introns=300)
    prefix = os.path.join(workdir, "twoseed")
    index_build.build_index_from_genome(g, prefix)
    ix = host.Index(prefix); orc = oracle_py.Oracle(prefix)
    gpu = host.DartGPU(ix, host.default_params(paired=1, max_mismatch=5))
    m1, m2 = synth.make_reads(g, 40000, rlen=101, seed=62, sub_rate=0.02, indel_frac=0.05, spliced_frac=0.15, n_frac=0.02)
    so, rl, flat = host.pack_reads(host.interleave_pairs(m1, m2))
    want = orc.map_batch(orc.params(paired=1, max_mismatch=5), so, rl, flat, threads=16)
    base = gpu.map_batch(so, rl, flat); base_ctr = gpu.counters()
    assert_same(base, want)
    assert base_ctr["seedq_trips_step"] > 0                # the queue kernel really ran
    monkeypatch.setenv("DG_SEED_PHASES", "1")
    gpu.set_params(gpu.params)                             # the DG_* switches are read at init and at dg_set_params, not per batch
    assert_same(gpu.map_batch(so, rl, flat), want)
    ctr = gpu.counters()
    assert ctr["seedq_trips_step"] > 0 and ctr["seedq_phases"] > 0
    for k in ("steps", "lf_steps", "sa_lookups", "seeds"):
        assert ctr[k] == base_ctr[k], k
    # Occ blocks: k_seed_qf finishes intervals of up to four rows by comparing their texts and counts those steps as one block each
    # (the reference loads a second block when the interval's two rows lie in different blocks): a lower bound, within a percent
    assert 0 <= ctr["occ_blocks"] - base_ctr["occ_blocks"] <= 0.01 * ctr["occ_blocks"]
    phased_ctr = ctr
    monkeypatch.setenv("DG_SEED_MULTI", "0")               # k_seed_qf with single-row comparisons only: block for block what the phased kernel counts
    monkeypatch.delenv("DG_SEED_PHASES")
    gpu.set_params(gpu.params)
    assert_same(gpu.map_batch(so, rl, flat), want)
    ctr = gpu.counters()
    for k in ("steps", "lf_steps", "sa_lookups", "seeds", "occ_blocks"):
        assert ctr[k] == phased_ctr[k], k
    assert ctr["seedq_trips_step"] > base_ctr["seedq_trips_step"]          # (the few-row intervals went through Occ steps again)
    monkeypatch.delenv("DG_SEED_MULTI")
    monkeypatch.setenv("DG_SEED_LEGACY", "1")
    gpu.set_params(gpu.params)
    assert_same(gpu.map_batch(so, rl, flat), want)
    ctr = gpu.counters()
    assert ctr["seedq_trips_step"] == 0
    for k in ("steps", "lf_steps", "sa_lookups", "seeds", "occ_blocks"):
        assert ctr[k] == phased_ctr[k], k
    monkeypatch.delenv("DG_SEED_LEGACY")
    gpu.close(); orc.close()
def test_gpu_seeds_at_the_ends_of_the_text(workdir):
    """reads from the first and last bases of the genome, both strands -- their matches end at the strand boundary (position l_pac of the
    text = forward strand + reverse complement) or at the end of the text, where the text comparisons of the seeding kernels take their
    slow path (d_text16_slow) -- and the same stretches planted a second and third time inside the genome, so that the few-row comparison of
    k_seed_qf holds rows next to a boundary together with ordinary ones"""
    g = synth.make_genome([400000, 300000], seed=91, repeat_scale=5.0, n_introns=0)
    L = g.total
    g.codes[150000:150300] = g.codes[:300]                  # the genome's first 300 bases again ...
    g.codes[500000:500300] = g.codes[L - 300:]              # ... and its last 300
    g.codes[250000:250300] = g.codes[:300]
    g.codes[399700:400000] = g.codes[L - 300:]              # the end of chromosome 1 = the end of the genome
    prefix = os.path.join(workdir, "ends")
    index_build.build_index_from_genome(g, prefix)
    ix = host.Index(prefix); orc = oracle_py.Oracle(prefix)
    asc = g.ascii()
    comp = np.zeros(256, np.uint8); comp[list(b"ACGT")] = list(b"TGCA")
    rng = np.random.default_rng(5)
    seqs = []
    for k in range(0, 120, 3):
        for p, ln in ((k, 101), (L - 101 - k, 101), (k, 60), (L - 60 - k, 60), (150000 + k, 101), (500000 + 199 - k, 101), (399700 + 199 - k, 101), (L - 150 - k, 150)):
            r = asc[p:p + ln].copy()
            if k % 2: r[int(rng.integers(20, ln - 20))] = ord("ACGT"[int(rng.integers(0, 4))])
            seqs.append(bytes(r.tobytes()))
            seqs.append(bytes(comp[r[::-1]].tobytes()))
    so, rl, flat = host.pack_reads(seqs)
    gpu = host.DartGPU(ix, host.default_params(paired=0, max_mismatch=5, multi_hit=1))
    # (-max_dup 1 / 2: the planted stretches occur three times, so an interval passes or fails the limit depending on how far the comparison narrows it)
    for kw in (dict(paired=0, max_mismatch=5, multi_hit=1), dict(paired=1, max_mismatch=3), dict(paired=0, max_mismatch=5, max_dup=2), dict(paired=0, max_mismatch=5, max_dup=1, multi_hit=1)):
        gpu.set_params(host.default_params(**kw))
        assert_same(gpu.map_batch(so, rl, flat), orc.map_batch(orc.params(**kw), so, rl, flat))
    gpu.set_params(host.default_params(paired=0, max_mismatch=5, multi_hit=1))
    want = gpu.probe_seeds(so, rl, flat)
    import os as _os
    for env in ({"DG_SEED_MULTI": "0"}, {"DG_SEED_LEGACY": "1"}, {"DG_SEED_PHASES": "1"}):
        for k_, v_ in env.items(): _os.environ[k_] = v_
        try:
            gpu.set_params(host.default_params(paired=0, max_mismatch=5, multi_hit=1))
            got = gpu.probe_seeds(so, rl, flat)
        finally:
            for k_ in env: del _os.environ[k_]
        for a, b in zip(want, got):
            assert np.array_equal(a, b), env
    gpu.close(); orc.close()


def test_gpu_index_aids_off_or_sampled(workdir, monkeypatch):
    """dg_init's index aids are chosen by text size (dg_api.hip: full suffix array up to 12 G symbols, every 2nd / 4th row beyond; prefix table
    K = 8..16): here they are forced to what a much larger genome would get -- no dense SA at all (LF walks to the reference's every-32nd-row
    samples), every 2nd / 4th row (LF steps, then the sample), no prefix table, a short one -- and every combination gives the oracle's records
    and the same reference-equivalent counters, through the queue kernel and the lane-per-read kernel"""
    g = synth.make_genome([1200000, 600000], seed=71, repeat_scale=40.0, n_introns=100)
    prefix = os.path.join(workdir, "aids")
    index_build.build_index_from_genome(g, prefix)
    ix = host.Index(prefix); orc = oracle_py.Oracle(prefix)
    m1, m2 = synth.make_reads(g, 12000, rlen=101, seed=72, sub_rate=0.02, indel_frac=0.05, spliced_frac=0.1, n_frac=0.02)
    so, rl, flat = host.pack_reads(host.interleave_pairs(m1, m2))
    want = orc.map_batch(orc.params(paired=1, max_mismatch=5), so, rl, flat, threads=16)
    base = None
    for dense, K, legacy in (("1", "", "0"), ("0", "", "0"), ("2", "", "0"), ("4", "10", "0"), ("1", "0", "0"), ("0", "0", "1"), ("4", "", "1")):
        monkeypatch.setenv("DG_SA_DENSE", dense); monkeypatch.setenv("DG_SEED_LEGACY", legacy)
        if K: monkeypatch.setenv("DG_KTAB_K", K)
        else: monkeypatch.delenv("DG_KTAB_K", raising=False)
        gpu = host.DartGPU(ix, host.default_params(paired=1, max_mismatch=5))
        assert_same(gpu.map_batch(so, rl, flat), want)
        c = gpu.counters()
        if base is None: base = c
        for k in ("steps", "lf_steps", "sa_lookups", "seeds"):
            assert c[k] == base[k], (k, dense, K, legacy)
        gpu.close()
    for k in ("DG_SA_DENSE", "DG_SEED_LEGACY"): monkeypatch.delenv(k)
    monkeypatch.delenv("DG_KTAB_K", raising=False)
    # -max_dup beyond what k_seed_qf's slot state counts (31 hits x max_dup in 20 bits): the phased queue kernel takes the batch
    gpu = host.DartGPU(ix, host.default_params(paired=1, max_mismatch=5, max_dup=40000))
    assert_same(gpu.map_batch(so[:6000], rl[:6000], flat), orc.map_batch(orc.params(paired=1, max_mismatch=5, max_dup=40000), so[:6000], rl[:6000], flat, threads=16))
    c = gpu.counters()
    assert c["seedq_trips_step"] > 0 and c["seedq_phases"] > 0
    gpu.close(); orc.close()


def test_gpu_start_up_paths_give_the_same_records(workdir):
    """dg_init (the index as host arrays) and dg_init_files (the index files straight to HBM, in-place Occ re-layout chunk by chunk), the look-up
    aids built before the call returns or by the library thread while batches already map (DG_INIT_ASYNC_AIDS: a context adopts each aid at
    its next batch, so the first batches run without them), full aids or the lean ones a short job gets (dg_index_files::expected_reads):
    every combination gives the oracle's records -- the aids change no result -- and clones created before the aids exist pick them up too."""
    g = synth.make_genome([1500000, 700000], seed=171, repeat_scale=30.0, n_introns=120)
    prefix = os.path.join(workdir, "startup")
    index_build.build_index_from_genome(g, prefix)
    ix = host.Index(prefix); orc = oracle_py.Oracle(prefix)
    m1, m2 = synth.make_reads(g, 15000, rlen=101, seed=172, sub_rate=0.02, indel_frac=0.05, spliced_frac=0.1, n_frac=0.01)
    so, rl, flat = host.pack_reads(host.interleave_pairs(m1, m2))
    want = orc.map_batch(orc.params(paired=1, max_mismatch=5), so, rl, flat, threads=16)
    p = host.default_params(paired=1, max_mismatch=5)
    for kw in (dict(from_files=False), dict(from_files=True), dict(from_files=True, expected_reads=30000),
               dict(from_files=True, async_aids=True), dict(from_files=True, async_aids=True, expected_reads=30000)):
        gpu = host.DartGPU(ix, p, **kw)
        cl = gpu.clone()                                        # (created while the aids may still be missing)
        for k in range(3):                                      # the first of these run beside the aid build
            assert_same(gpu.map_batch(so, rl, flat), want)
            assert_same(cl.map_batch(so, rl, flat), want)
        gpu.wait_index()
        assert_same(gpu.map_batch(so, rl, flat), want); assert_same(cl.map_batch(so, rl, flat), want)
        rep = gpu.init_report()
        assert ("lean aids" in rep) == ("expected_reads" in kw) and ("beside the first batches" in rep) == bool(kw.get("async_aids")), (kw, rep)
        assert "k_build_ktab" in rep and ("host arrays" in rep) == (not kw["from_files"]), rep
        gpu.close()
    # index files that are missing or shorter than a header: an argument error with a message (the host program prints the reference's
    # "Index files are corrupt"), not a crash
    import ctypes as C
    lib = host._load_lib()
    f = ix.files()
    for bad in (prefix + ".nothere", None):
        if bad is None:
            bad = prefix + "_short.sa"
            open(bad, "wb").write(b"x" * 20)
        f.sa_path = bad.encode()
        st = C.c_int(0)
        assert not lib.dg_init_files(C.byref(f), C.byref(p), 0, 0, C.byref(st)) and st.value == -3 and b"cannot read" in lib.dg_last_error(None)
    orc.close()


def test_gpu_packed_reads_and_pinned_buffers(workdir):
    """dg_map_batch_packed (2 bit/base + N list) gives the records of dg_map_batch on the same reads, with fixed and with
    per-read lengths; page-locked caller buffers (dg_host_alloc) through the raw ABI; a read with a lower-case base is refused
    by the packer (it must go through the ASCII entry)"""
    import ctypes as C
    g = synth.make_genome([1500000, 500000], seed=81, repeat_scale=60.0, n_introns=150)
    prefix = os.path.join(workdir, "packed")
    index_build.build_index_from_genome(g, prefix)
    ix = host.Index(prefix); orc = oracle_py.Oracle(prefix)
    gpu = host.DartGPU(ix, host.default_params(paired=1, max_mismatch=5))
    m1, m2 = synth.make_reads(g, 25000, rlen=101, seed=82, sub_rate=0.015, indel_frac=0.04, spliced_frac=0.1, n_frac=0.01)
    arr = host.interleave_pairs(m1, m2)
    so, rl, flat = host.pack_reads(arr)
    want = orc.map_batch(orc.params(paired=1, max_mismatch=5), so, rl, flat, threads=16)
    words, nlist = host.pack_reads_2bit(arr)
    assert len(nlist) > 300
    assert_same(gpu.map_batch_packed(words, nlist, 101), want)
    res_c = gpu.download_compact()                       # the same records through the 12 + 16 byte types
    assert_same(res_c, want)
    assert 0 < gpu.last_compact_ops < len(want[2])      # plain "101M" reports travel without their CIGAR op
    assert_same(gpu.map_batch_compact(words, nlist, 101), want)      # dg_map_batch_compact: one call, records packed inside the run, arrays grown on DG_ERR_CAPACITY
    # a compact-only call leaves the units k_pair finished without the full record types (round 5); a caller that asks for them after all (DG_ERR_RANGE's
    # way out) gets the batch mapped again, with them, by dg_batch_download itself
    f_reads = np.zeros(len(want[0]), host.READ_OUT); f_rep = np.zeros(len(want[1]) + 16, host.REPORT_OUT); f_cig = np.zeros(len(want[2]) + 16, np.uint32); f_sj = np.zeros(len(want[3]) + 16, host.SJ_OUT)
    f_caps = (C.c_size_t * 3)(len(f_rep), len(f_cig), len(f_sj))
    assert gpu.lib.dg_batch_download(gpu.ctx, f_reads.ctypes.data, f_rep.ctypes.data, f_cig.ctypes.data, f_sj.ctypes.data, f_caps) == 0, gpu.lib.dg_last_error(gpu.ctx)
    assert_same(host.BatchResult(f_reads, f_rep[:len(want[1])], f_cig[:len(want[2])], f_sj[:len(want[3])]), want)
    # ragged: every read cut to its own length (the tail bases stay in the words, the lengths say where the read ends)
    rng = np.random.default_rng(5)
    lens = rng.integers(30, 102, size=len(arr)).astype(np.uint16)
    seqs = [arr[i, :lens[i]].tobytes() for i in range(len(arr))]
    so2, rl2, flat2 = host.pack_reads(seqs)
    keep = np.nonzero(arr == ord("N"))
    inside = keep[1] < lens[keep[0]]
    nl2 = (keep[0][inside].astype(np.uint64) * words.shape[1] * 16 + keep[1][inside].astype(np.uint64)).astype(np.uint32)
    lens[0] = 101                                        # words_per_read is ceil(longest / 16)
    seqs[0] = arr[0].tobytes(); so2, rl2, flat2 = host.pack_reads(seqs)
    keep = np.nonzero(arr == ord("N")); inside = keep[1] < lens[keep[0]]
    nl2 = (keep[0][inside].astype(np.uint64) * words.shape[1] * 16 + keep[1][inside].astype(np.uint64)).astype(np.uint32)
    want2 = orc.map_batch(orc.params(paired=1, max_mismatch=5), so2, rl2, flat2, threads=16)
    assert_same(gpu.map_batch_packed(words, nl2, 0, rlen=lens), want2)
    assert_same(gpu.download_compact(), want2)           # "<length of the read>M" comes back per read
    assert_same(gpu.map_batch_compact(words, nl2, 0, rlen=lens), want2)
    low = arr[:4].copy(); low[1, 7] = ord("a")
    with pytest.raises(ValueError):
        host.pack_reads_2bit(low)
    # pinned buffers through dg_map_batch itself
    n = len(rl)
    p_so = gpu.pinned((n,), np.uint32); p_rl = gpu.pinned((n,), np.uint16); p_seq = gpu.pinned((len(flat) + 64,), np.uint8)
    p_so.a[:] = so; p_rl.a[:] = rl; p_seq.a[:len(flat)] = flat
    caps = (C.c_size_t * 3)(n * 8, n * 16, n * 2); used = (C.c_size_t * 3)()     # (a repeat-rich genome: ~5 reports per read)
    o_r = gpu.pinned((n,), host.READ_OUT); o_p = gpu.pinned((caps[0],), host.REPORT_OUT); o_c = gpu.pinned((caps[1],), np.uint32); o_s = gpu.pinned((caps[2],), host.SJ_OUT)
    rc = gpu.lib.dg_map_batch(gpu.ctx, n, p_so.a.ctypes.data, p_rl.a.ctypes.data, p_seq.a.ctypes.data, o_r.a.ctypes.data, o_p.a.ctypes.data, o_c.a.ctypes.data, o_s.a.ctypes.data, caps, used)
    assert rc == 0, (rc, gpu.lib.dg_last_error(gpu.ctx), list(used), list(caps))
    assert_same(host.BatchResult(o_r.a.copy(), o_p.a[:used[0]].copy(), o_c.a[:used[1]].copy(), o_s.a[:used[2]].copy()), want)
    # dg_map_batch writes the full records only: asking for the compact ones afterwards is an argument error with a message, not stale data
    n_ops = C.c_size_t(0)
    rc = gpu.lib.dg_batch_download_compact(gpu.ctx, o_r.a.ctypes.data, o_p.a.ctypes.data, o_c.a.ctypes.data, o_s.a.ctypes.data, caps, C.byref(n_ops))
    assert rc == -3 and b"full records only" in gpu.lib.dg_last_error(gpu.ctx), (rc, gpu.lib.dg_last_error(gpu.ctx))
    # the same call with ordinary (not page-locked) arrays: the download then goes through the context's own stream, same records
    q_r = np.zeros(n, host.READ_OUT); q_p = np.zeros(caps[0], host.REPORT_OUT); q_c = np.zeros(caps[1], np.uint32); q_s = np.zeros(caps[2], host.SJ_OUT)
    rc = gpu.lib.dg_map_batch(gpu.ctx, n, p_so.a.ctypes.data, p_rl.a.ctypes.data, p_seq.a.ctypes.data, q_r.ctypes.data, q_p.ctypes.data, q_c.ctypes.data, q_s.ctypes.data, caps, used)
    assert rc == 0, (rc, gpu.lib.dg_last_error(gpu.ctx))
    assert_same(host.BatchResult(q_r, q_p[:used[0]], q_c[:used[1]], q_s[:used[2]]), want)
    for p in (p_so, p_rl, p_seq, o_r, o_p, o_c, o_s):
        p.free()
    gpu.close(); orc.close()


def test_gpu_capacity_estimates_grow_and_results_stay(workdir):
    """the data-dependent buffers are sized from estimates and grown on overflow (no mid-batch size read-back): a first batch
    far denser in seeds / reports than the estimates (repeat family reads, -m, -max_dup 1000) and a sparse one afterwards both
    equal the oracle"""
    g = synth.make_genome([800000], seed=91, repeat_scale=400.0)
    prefix = os.path.join(workdir, "caps")
    index_build.build_index_from_genome(g, prefix)
    ix = host.Index(prefix); orc = oracle_py.Oracle(prefix)
    p, _ = common.parse_flags(["-mis", "8", "-m", "-max_dup", "1000"])
    gpu = host.DartGPU(ix, host.default_params(paired=1, **p))
    m1, m2 = synth.make_reads(g, 3000, rlen=151, seed=92, sub_rate=0.02, indel_frac=0.1)
    so, rl, flat = host.pack_reads(host.interleave_pairs(m1, m2))
    want = orc.map_batch(orc.params(paired=1, **p), so, rl, flat, threads=16)
    assert len(want[1]) > 3 * len(rl)                    # many reports per read: the first estimate (1.25 per read) is too small
    assert_same(gpu.map_batch(so, rl, flat), want)
    assert_same(gpu.map_batch(so[:200], rl[:200], flat), orc.map_batch(orc.params(paired=1, **p), so[:200], rl[:200], flat, threads=4))
    gpu.close(); orc.close()


def test_gpu_compact_records_range_and_wide_chromosome_table(workdir):
    """66 000 chromosomes: the chromosome index does not fit k_pair's report slots (every unit takes the general path) nor the
    compact record types (dg_batch_download_compact answers DG_ERR_RANGE, the full records are right)"""
    g = synth.make_genome([120] * 66000, seed=5, repeat_scale=0.0)
    prefix = os.path.join(workdir, "manychr")
    index_build.build_index_from_genome(g, prefix)
    ix = host.Index(prefix); orc = oracle_py.Oracle(prefix)
    gpu = host.DartGPU(ix, host.default_params(paired=0, max_mismatch=3))
    s1, _ = synth.make_reads(g, 6000, rlen=60, seed=6, paired=False, indel_frac=0.0, n_frac=0.0)
    so, rl, flat = host.pack_reads(s1)
    want = orc.map_batch(orc.params(paired=0, max_mismatch=3), so, rl, flat, threads=8)
    assert int(want[1]["chr"].max()) > 0xFFFF
    res = gpu.map_batch(so, rl, flat)
    assert_same(res, want)
    assert gpu.counters()["general_path_units"] == len(rl)
    with pytest.raises(RuntimeError) as e:
        gpu.download_compact()
    assert "(-6)" in str(e.value)
    assert_same(gpu.map_batch(so[:2000], rl[:2000], flat), orc.map_batch(orc.params(paired=0, max_mismatch=3), so[:2000], rl[:2000], flat, threads=8))
    gpu.close(); orc.close()


def test_gpu_random_parity_sweep(workdir):
    """ten random configurations (tests/probes/fuzz_parity.py: fresh genome of 1-4 chromosomes with 0-100 x repeat families, read
    length 36-250, single or paired, substitution / indel / splice / N rates, every flag at random): all records against the oracle,
    through the ASCII, the packed and the compact entry points.  (The probe itself takes any number of rounds and any first seed:
    1 714 other rounds were identical by the end of round 2.)"""
    import sys
    sys.path.insert(0, os.path.join(common.ROOT, "tests", "probes"))
    import fuzz_parity
    assert fuzz_parity.run(10, 9000, workdir=os.path.join(workdir, "fuzz"), log=lambda m: None) == 10



def test_gpu_scan_stress_many_small_batches_on_twelve_contexts(workdir):
    """The single-pass scans (dg_scan.h) under the conditions in which round 2 once saw a look-back give up: twelve contexts in flight, every
    one mapping small batches back to back (2 400 batches in all; batch sizes cycle so that a run's tiles meet the previous run's words at
    every position).  No batch may have been run again because of a scan (`reruns_scan_total` == 0: the state words carry their run's
    epoch and are never zeroed) and every context's last records equal the oracle's."""
    import threading
    g = synth.make_genome([600000, 400000], seed=71, repeat_scale=30.0, n_introns=100)
    prefix = os.path.join(workdir, "stress")
    index_build.build_index_from_genome(g, prefix)
    ix = host.Index(prefix); orc = oracle_py.Oracle(prefix)
    gpu = host.DartGPU(ix, host.default_params(paired=1, max_mismatch=5))
    ctx = [gpu] + [gpu.clone() for _ in range(11)]
    sizes = (3000, 1100, 2600, 700)
    batches = []
    for j, n in enumerate(sizes):
        m1, m2 = synth.make_reads(g, n, rlen=101, seed=72 + j, spliced_frac=0.1, indel_frac=0.04, n_frac=0.01)
        arr = host.interleave_pairs(m1, m2)
        batches.append((host.pack_reads(arr), host.pack_reads_2bit(arr)))
    want = [orc.map_batch(orc.params(paired=1, max_mismatch=5), *b[0], threads=16) for b in batches]
    out, errs = [None] * len(ctx), []
    def work(k):
        try:
            for i in range(200):
                j = (i + k) % len(sizes)
                words, nlist = batches[j][1]
                res = ctx[k].map_batch_compact(words, nlist, 101) if i % 2 else ctx[k].map_batch(*batches[j][0])
                out[k] = (j, res)
        except Exception as e:
            errs.append(e)
    th = [threading.Thread(target=work, args=(k,)) for k in range(len(ctx))]
    for t in th: t.start()
    for t in th: t.join()
    assert not errs, errs
    for k in range(len(ctx)):
        j, res = out[k]
        assert_same(res, want[j])
        assert ctx[k].counters()["reruns_scan_total"] == 0, k
    gpu.close(); orc.close()


def test_gpu_scan_timeout_rerun_path(workdir, monkeypatch):
    """The path a look-back that runs out of its poll budget takes, forced: DG_SCAN_POLL_BUDGET=1 gives the first attempt of every batch
    a budget of one poll, so some tile gives up (DG_E_SCAN), the host records what the poller saw and runs the batch again with the
    normal budget; the records are the oracle's and the re-run is counted."""
    g = synth.make_genome([900000], seed=81, repeat_scale=30.0)
    prefix = os.path.join(workdir, "scanrerun")
    index_build.build_index_from_genome(g, prefix)
    ix = host.Index(prefix); orc = oracle_py.Oracle(prefix)
    m1, m2 = synth.make_reads(g, 60000, rlen=101, seed=82, indel_frac=0.04, n_frac=0.01)
    so, rl, flat = host.pack_reads(host.interleave_pairs(m1, m2))
    want = orc.map_batch(orc.params(paired=1, max_mismatch=5), so, rl, flat, threads=16)
    gpu = host.DartGPU(ix, host.default_params(paired=1, max_mismatch=5))
    assert_same(gpu.map_batch(so, rl, flat), want)         # (sizes the context's buffers: a capacity re-run would hide the scan's)
    # each of the batch's three single-pass scans on its own (DG_SCAN_POLL_SCANS: 1 = k_seed_offsets, 2 = k_pair, 4 = k_emit_slow), then all three:
    # a give-up in k_seed_offsets leaves seed_off[] unwritten for the tiles behind it -- k_pair must not touch them (it leaves at entry)
    for mask in ("1", "2", "4", "7"):
        monkeypatch.setenv("DG_SCAN_POLL_BUDGET", "1"); monkeypatch.setenv("DG_SCAN_POLL_SCANS", mask)
        gpu.set_params(gpu.params)                         # the DG_* switches are read at init and at dg_set_params
        before = gpu.counters()["reruns_scan_total"]
        res = gpu.map_batch(so, rl, flat)
        c = gpu.counters()
        monkeypatch.delenv("DG_SCAN_POLL_BUDGET"); monkeypatch.delenv("DG_SCAN_POLL_SCANS")
        gpu.set_params(gpu.params)
        assert_same(res, want)
        assert c["reruns_scan_total"] == before + 1 and c["batch_runs"] == 2, (mask, c)
        msg = gpu.lib.dg_last_error(gpu.ctx) or b""
        assert b"look-back" in msg and b"stuck tile's trace" in msg, msg
    gpu.close(); orc.close()


def test_gpu_ecoli_sized_single_end_config0(workdir):
    """BASELINE configs[0] at its size on the HIP path: a 4 641 652 bp one-chromosome genome, 100 k single-end 100 bp reads, the reference's
    default flags (MaxMismatch 0) and -mis 5, every record against the oracle."""
    g = synth.make_genome([4641652], seed=20, repeat_scale=1.0, names=["ecoli"])
    prefix = os.path.join(workdir, "ecoli")
    index_build.build_index_from_genome(g, prefix)
    ix = host.Index(prefix); orc = oracle_py.Oracle(prefix)
    m1, _ = synth.make_reads(g, 100000, rlen=100, seed=21, sub_rate=0.01, indel_frac=0.02, n_frac=0.002, paired=False)
    so, rl, flat = host.pack_reads(m1)
    gpu = host.DartGPU(ix, host.default_params(paired=0))
    for mis in (0, 5):
        gpu.set_params(host.default_params(paired=0, max_mismatch=mis))
        assert_same(gpu.map_batch(so, rl, flat), orc.map_batch(orc.params(paired=0, max_mismatch=mis), so, rl, flat, threads=16))
    gpu.close(); orc.close()


def test_gpu_packed_nlist_out_of_range_is_an_argument_error(workdir):
    """a packed batch whose N list points outside the batch (a buggy packer) is refused with DG_ERR_ARG instead of writing outside the buffers"""
    c = common.build_case("se100", workdir)
    ix = host.Index(c["prefix"])
    gpu = host.DartGPU(ix, host.default_params(paired=0, max_mismatch=5))
    arr = np.ascontiguousarray(c["reads"][:64]).copy()
    arr[arr == ord("N")] = ord("A")
    rl0 = arr.shape[1]
    words, nlist = host.pack_reads_2bit(arr)
    bad = np.array([5, 64 * 16 * words.shape[1] + 7, 0xFFFFFFF0], np.uint32)
    gpu.upload_packed(words, bad, rl0)
    used = (host.C.c_size_t * 3)()
    assert gpu.lib.dg_batch_run(gpu.ctx, used) == -3
    assert b"N list" in gpu.lib.dg_last_error(gpu.ctx)
    good = gpu.map_batch_packed(words, np.array([5], np.uint32), rl0)       # the context is usable afterwards
    arr[0, 5] = ord("N")
    ref = gpu.map_batch(*host.pack_reads(arr))
    assert_same(good, (ref.reads, ref.reports, ref.cigar, ref.sj))
    gpu.close()
