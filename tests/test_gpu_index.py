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


@pytest.mark.parametrize("sorter", ["plain", "bucketed"])
@pytest.mark.parametrize("name", CASES)
def test_gpu_index_builder_matches_reference_indexer(name, sorter, workdir, monkeypatch):
    import torch
    assert torch.cuda.is_available()
    spec = common.MANIFEST["cases"][name]
    g = synth.make_genome(spec["lengths"], seed=spec["gseed"], repeat_scale=spec["rscale"], n_introns=spec["nintr"])
    if sorter == "bucketed":
        monkeypatch.setenv("DART_SA_BUCKETED", "1")
    else:
        monkeypatch.delenv("DART_SA_BUCKETED", raising=False)
    prefix = os.path.join(workdir, "gpuidx_%s_%s" % (name, sorter))
    index_build.build_index_from_genome(g, prefix, device="cuda")
    for ext, want in common.MANIFEST["manifest"][name]["index_sha256"].items():
        assert common.sha(prefix + "." + ext) == want, "GPU-built .%s differs from the reference bwt_index output (%s sorter)" % (ext, sorter)


@pytest.mark.parametrize("sorter", ["plain", "bucketed"])
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
    if sorter == "bucketed":
        monkeypatch.setenv("DART_SA_BUCKETED", "1")
    else:
        monkeypatch.delenv("DART_SA_BUCKETED", raising=False)
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
