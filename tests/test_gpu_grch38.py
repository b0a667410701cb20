"""GPU suite (-m gpu) at GRCh38 index size: BASELINE configs[2] ("Full GRCh38 index, 2x101") and configs[4] ("Full GRCh38,
2x151 with -max_intron 500000") against the oracle, on a GRCh38-SIZED synthetic genome (24 chromosomes with the GRCh38 lengths,
3 088 269 832 bp, 20 000 planted introns; the real sequence cannot be fetched offline).  This is the size at which the
K = 16 prefix table, the 40-bit interval fields, the 50 GB full suffix array and the grid-stride init kernels are used at all;
the chr20-sized tests do not reach them.  One module-scoped fixture builds the index once (GPU suffix sort, ~1-2 min, cached
under /tmp) and keeps one device context + one oracle instance for all tests."""
import os
import numpy as np
import pytest
import oracle_py
from dart_amd import host, synth
from test_gpu_parity import assert_same

pytestmark = pytest.mark.gpu
N_INTRONS = 20000


@pytest.fixture(scope="module")
def big():
    import bench
    cache = os.environ.get("DART_BENCH_CACHE", "/tmp/dart_bench_cache")
    label, names, lens = bench.genome_spec("grch38")
    prefix, g = bench.prepare_index(cache, (names, lens), 0, lambda: None, N_INTRONS)
    ix = host.Index(prefix)
    assert ix.seq_len == 2 * sum(lens) and ix.seq_len > (1 << 32)            # positions need more than 32 bits
    gpu = host.DartGPU(ix, host.default_params(paired=1, max_mismatch=5))
    orc = oracle_py.Oracle(prefix)
    yield g, ix, gpu, orc
    gpu.close(); orc.close()


def _batch(g, n_pairs, rlen, seed, spliced=0.0):
    m1, m2 = synth.make_reads(g, n_pairs, rlen=rlen, seed=seed, sub_rate=0.01, indel_frac=0.02, n_frac=0.002, spliced_frac=spliced)
    return host.pack_reads(host.interleave_pairs(m1, m2))


def test_grch38_sized_2x101_matches_oracle(big):
    """configs[2] shape: 200 k pairs 2x101, `-mis 5` and the reference's default flags; every record field, CIGAR op and
    splice-junction tuple, plus the reference-equivalent work counters (SURVEY 8d accounting)"""
    g, ix, gpu, orc = big
    so, rl, flat = _batch(g, 200000, 101, 1000)
    for mis in (5, 0):
        gpu.set_params(host.default_params(paired=1, max_mismatch=mis))
        res = gpu.map_batch(so, rl, flat)
        assert_same(res, orc.map_batch(orc.params(paired=1, max_mismatch=mis), so, rl, flat, threads=16))
        c, oc = gpu.counters(), orc.counters
        assert c["steps"] == oc["n_2occ4"] and c["lf_steps"] == oc["n_lf"] and c["sa_lookups"] == oc["n_sa"]
        assert c["nw_calls"] == oc["n_nw"] and c["nw_cells"] == oc["nw_cells"]
        assert 0 <= oc["n_occ_blocks"] - (c["occ_blocks"] + c["lf_steps"]) <= 1e-2 * oc["n_occ_blocks"]
    assert float((res.reads["score"] > 0).mean()) > 0.5


def test_grch38_sized_spliced_2x151_matches_oracle(big):
    """configs[4] shape at its real index size: 100 k pairs 2x151, 30 % of the reads span a planted intron of 200 b - 500 kb,
    -max_intron 500000 (long-gap re-seeding windows + gap-filling alignments)"""
    g, ix, gpu, orc = big
    so, rl, flat = _batch(g, 100000, 151, 1001, spliced=0.3)
    gpu.set_params(host.default_params(paired=1, max_mismatch=5, max_intron=500000))
    res = gpu.map_batch(so, rl, flat)
    assert_same(res, orc.map_batch(orc.params(paired=1, max_mismatch=5, max_intron=500000), so, rl, flat, threads=16))
    assert len(res.sj) > 3000 and gpu.counters()["reseed_calls"] > 100       # the junction and re-seeding paths really ran
    assert gpu.counters()["reseed_calls"] == orc.counters["n_reseed"] and gpu.counters()["reseed_window"] == orc.counters["reseed_window"]


def test_grch38_sized_full_batch_split_invariance(big):
    """a full 1 M-pair batch (the bench's unit of work) against the same reads mapped as three uneven batches, and against the
    oracle on a slice from its middle: per-read records do not depend on how the reads are batched"""
    g, ix, gpu, orc = big
    gpu.set_params(host.default_params(paired=1, max_mismatch=5))
    m1, m2 = synth.make_reads(g, 1000000, rlen=101, seed=1002, sub_rate=0.01, indel_frac=0.02, n_frac=0.002)
    arr = host.interleave_pairs(m1, m2)
    whole = gpu.map_batch(*host.pack_reads(arr))
    cuts = [0, 2 * 310001, 2 * 777777, 2 * 1000000]
    parts = [gpu.map_batch(*host.pack_reads(arr[a:b])) for a, b in zip(cuts[:-1], cuts[1:])]
    for f in ("score", "sub_score", "mis_num", "mapq", "n_rep", "best", "n_sj"):
        assert np.array_equal(np.concatenate([p.reads[f] for p in parts]), whole.reads[f]), f
    for f in ("aln_score", "sj_type", "flag", "paired_idx", "chr", "bdir", "pos", "n_cigar"):
        assert np.array_equal(np.concatenate([p.reports[f] for p in parts]), whole.reports[f]), f
    cig = lambda r: [r.cigar[o:o + k].tobytes() for o, k in zip(r.reports["cigar_off"], r.reports["n_cigar"])]
    assert sum((cig(p) for p in parts), []) == cig(whole)
    lo, hi = 2 * 500000, 2 * 540000
    o_reads, o_rep, o_cig, o_sj = orc.map_batch(orc.params(paired=1, max_mismatch=5), *host.pack_reads(arr[lo:hi]), threads=16)
    for f in ("score", "sub_score", "mis_num", "mapq", "n_rep", "best", "n_sj"):
        assert np.array_equal(o_reads[f], whole.reads[f][lo:hi]), f
    r0 = int(whole.reads["rep_off"][lo])
    for f in ("aln_score", "sj_type", "flag", "paired_idx", "chr", "bdir", "pos", "n_cigar"):
        assert np.array_equal(o_rep[f], whole.reports[f][r0:r0 + len(o_rep)]), f
