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
