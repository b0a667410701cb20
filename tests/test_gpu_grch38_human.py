"""GPU suite (-m gpu): a GRCh38-SIZED genome with HUMAN-LIKE repeat content (dart_amd/synth.py::_make_genome_human: about half of the
3.09 Gbp in SINE / LINE / older interspersed families, segmental duplications, satellite arrays and microsatellites) against the oracle.
The i.i.d. + planted-repeats genome of the other GRCh38-sized tests leaves the paths real DNA stresses almost idle: searches that
restart base by base because `freq == 0` (AlignmentCandidates.cpp:209), intervals above MaxDupNum (bwt_search.cpp:173), reads with
dozens of candidates (Mapping.cpp:403-450), k_seed_heavy / k_chain_heavy / the general report path.  Own module: its index and its
123 GB of device tables must not live beside the other module's."""
import os
import numpy as np
import pytest
import oracle_py
from dart_amd import host, synth
from test_gpu_parity import assert_same

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def human():
    import bench
    cache = os.environ.get("DART_BENCH_CACHE", "/tmp/dart_bench_cache")
    label, names, lens = bench.genome_spec("grch38")
    prefix, g = bench.prepare_index(cache, (names, lens), 0, lambda: None, 0, 1.0, "human")
    ix = host.Index(prefix)
    assert ix.seq_len == 2 * sum(lens) and ix.seq_len > (1 << 32)
    gpu = host.DartGPU(ix, host.default_params(paired=1, max_mismatch=5))
    orc = oracle_py.Oracle(prefix)
    yield g, ix, gpu, orc
    gpu.close(); orc.close()


def test_grch38_sized_human_like_2x101_matches_oracle(human):
    """200 k pairs 2x101, `-mis 5` and the reference's default flags: every record field, CIGAR op and junction tuple, and the
    reference-equivalent work counters; the repeat paths must really have run (units chained by a wave each, reads finished by
    k_seed_heavy's 64-wide speculation, the general report path)"""
    g, ix, gpu, orc = human
    m1, m2 = synth.make_reads(g, 200000, rlen=101, seed=1000, sub_rate=0.01, indel_frac=0.02, n_frac=0.002)
    so, rl, flat = host.pack_reads(host.interleave_pairs(m1, m2))
    for mis in (5, 0):
        gpu.set_params(host.default_params(paired=1, max_mismatch=mis))
        res = gpu.map_batch(so, rl, flat)
        assert_same(res, orc.map_batch(orc.params(paired=1, max_mismatch=mis), so, rl, flat, threads=16))
        c, oc = gpu.counters(), orc.counters
        assert c["steps"] == oc["n_2occ4"] and c["lf_steps"] == oc["n_lf"] and c["sa_lookups"] == oc["n_sa"]
        assert c["nw_calls"] == oc["n_nw"] and c["nw_cells"] == oc["nw_cells"]
        assert c["reseed_calls"] == oc["n_reseed"] and c["reseed_window"] == oc["reseed_window"]
        assert 0 <= oc["n_occ_blocks"] - (c["occ_blocks"] + c["lf_steps"]) <= 1e-2 * oc["n_occ_blocks"]
    assert c["wave_chained_units"] > 200 and c["general_path_units"] > 5000, c
    assert float((res.reads["n_rep"] > 1).mean()) > 0.05                     # reads with several candidates are common here
