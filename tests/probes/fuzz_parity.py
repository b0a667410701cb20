"""Random parity sweep (GPU box): fresh genomes, read sets and flag sets, every record against the oracle.
Usage: python tests/probes/fuzz_parity.py [rounds] [first seed]   -- prints one line per round, stops at the first difference."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import common, oracle_py
from dart_amd import synth, index_build, host
def run(rounds, seed0, workdir="/tmp/fuzz", log=print):
    """-> number of rounds that were identical; raises AssertionError at the first difference"""
    oracle_py.build()
    os.makedirs(workdir, exist_ok=True)
    for k in range(rounds):
        rng = np.random.default_rng(seed0 + k)
        lens = [int(x) for x in rng.integers(200000, 6000000, size=int(rng.integers(1, 5)))]
        g = synth.make_genome(lens, seed=seed0 + k, repeat_scale=float(rng.choice([0.0, 1.0, 20.0, 100.0])), n_introns=int(rng.choice([0, 200, 2000])))
        prefix = os.path.join(workdir, "g%d" % k)
        index_build.build_index_from_genome(g, prefix)
        ix = host.Index(prefix); orc = oracle_py.Oracle(prefix)
        rlen = int(rng.choice([36, 75, 101, 151, 250]))
        paired = int(rng.integers(0, 2))
        n = int(rng.choice([20000, 60000]))
        kw = dict(rlen=rlen, seed=seed0 + 7 * k, sub_rate=float(rng.choice([0.0, 0.01, 0.03])), indel_frac=float(rng.choice([0.0, 0.05, 0.3])),
                  spliced_frac=float(rng.choice([0.0, 0.2, 0.5])) if len(g.introns) and rlen >= 60 else 0.0, n_frac=float(rng.choice([0.0, 0.01, 0.05])))
        if paired:
            m1, m2 = synth.make_reads(g, n, **kw); arr = host.interleave_pairs(m1, m2)
        else:
            arr, _ = synth.make_reads(g, n, paired=False, **kw)
        flags = dict(max_mismatch=int(rng.choice([0, 2, 5, 10])), multi_hit=int(rng.integers(0, 2)), all_sj=int(rng.integers(0, 2)),
                     max_dup=int(rng.choice([100, 1000])), max_intron=int(rng.choice([500000, 50000])), min_intron=int(rng.choice([5, 20])))
        if k % 4 == 3:       # every fourth round: characters the reference's string code treats in its own way ('-' is a gap to AddNewCigarElements, case and IUPAC letters never equal the genome's)
            rate = float(rng.choice([0.002, 0.01]))
            arr = np.where(rng.random(arr.shape) < rate, rng.choice(np.frombuffer(b"---acgtRYn", np.uint8), size=arr.shape), arr).astype(np.uint8)
            kw = dict(kw, odd_characters=rate); flags["max_mismatch"] = int(rng.choice([5, 12, 30]))
        so, rl, flat = host.pack_reads(arr)
        gpu = host.DartGPU(ix, host.default_params(paired=paired, **flags))
        t = time.time()
        want = orc.map_batch(orc.params(paired=paired, **flags), so, rl, flat, threads=16)
        res = gpu.map_batch(so, rl, flat)
        try:
            common.assert_same(res, want)
            words_ok = ""
            if not (arr == ord("N")).all() and set(np.unique(arr).tolist()) <= set(b"ACGTN"):
                words, nlist = host.pack_reads_2bit(arr)
                common.assert_same(gpu.map_batch_packed(words, nlist, rlen), want)
                common.assert_same(gpu.download_compact(), want)
                common.assert_same(gpu.map_batch_compact(words, nlist, rlen), want)      # one call: compact records built inside the run
                words_ok = " + packed/compact"
            log("round %d ok%s: genome %s rscale, %d x %s%d, %s, %s  (%.1f s)" % (k, words_ok, lens, n, "2x" if paired else "", rlen, kw, flags, time.time() - t))
        except AssertionError as e:
            raise AssertionError("round %d (seed %d) DIFFERS: genome %s, %d x %s%d, %s, %s: %s" % (k, seed0 + k, lens, n, "2x" if paired else "", rlen, kw, flags, str(e)[:300]))
        gpu.close(); orc.close()
    return rounds


if __name__ == "__main__":
    n_rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    run(n_rounds, first, log=lambda m: print(m, flush=True))
    print("all %d rounds identical" % n_rounds)
