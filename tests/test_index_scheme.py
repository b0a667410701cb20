"""CPU suite: the suffix-sorting scheme of the MI355X index builder (dart_amd/index_build.py::suffix_array_hip -- buckets by two symbols,
a 31-symbol round-0 key whose low bits say how many symbols exist, rank pairs refined in place) run with tests/index_emul.py's numpy
restatement of the kernels' contracts, against the plain torch.sort prefix doubler whose files are pinned to the reference indexer's
digests (tests/test_oracle_golden.py).  What is covered here is the host logic and the scheme; the kernels are covered on the GPU."""
import numpy as np
import pytest
import torch
from dart_amd import index_build, synth
import index_emul


@pytest.mark.parametrize("seed,lengths,rscale", [(1, [3000, 1000], 1.0), (2, [50000], 20.0), (3, [150000, 60000], 20.0)])
@pytest.mark.parametrize("end", ["as generated", "ends in a run of A", "ends in A, starts with A"])
def test_suffix_array_scheme_of_the_hip_builder(seed, lengths, rscale, end):
    g = synth.make_genome(lengths, seed=seed, repeat_scale=rscale)
    f = g.codes.copy()
    if end == "ends in a run of A":
        f[:40] = 3                   # the reverse complement, hence the text, ends in 40 A: many suffixes meet the '$' inside one padded key
    elif end == "ends in A, starts with A":
        f[:3] = 3
        f[-5:] = 0
    n = 2 * len(f)
    sa, rank = index_build.suffix_array_hip(index_emul.pack_text(f), n, 3 - int(f[0]), ops=index_emul.EmulOps())
    want = index_build.suffix_array(torch.from_numpy(np.concatenate([f, (3 - f)[::-1]])))
    assert torch.equal(sa, want)
    assert torch.equal(rank[sa], torch.arange(n + 1))
