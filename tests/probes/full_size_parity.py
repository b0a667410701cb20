"""Usage: python tests/probes/full_size_parity.py [cfg3|cfg5|both] [batches]
SURVEY 8d's parity chain (2) at FULL size on the GRCh38-sized index: every record of the bench's own batches against the oracle.
  cfg3   the ten 1 M-pair batches of bench.py's step (seeds 1000 .. 1009: 20 M reads of 2x101), through the timed entry point
         (packed reads in, compact records out) -- BASELINE configs[2]
  cfg5   1 M pairs of 2x151 with 30 % of the reads spliced over planted introns, -max_intron 500000 -- BASELINE configs[4]'s shape
The oracle needs ~85 s of 16 cores per 20 M reads.  Prints one line per batch and a summary; exit code 1 on any difference."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench, oracle_py, common
from dart_amd import host, synth

what = sys.argv[1] if len(sys.argv) > 1 else "both"
n_batches = int(sys.argv[2]) if len(sys.argv) > 2 else 10
cores = bench.host_cores()
label, gnames, glens = bench.genome_spec("grch38")
bad_total = 0


def run(tag, prefix, g, rlen, spliced, max_intron, seeds, pairs):
    global bad_total
    ix = host.Index(prefix)
    gpu = host.DartGPU(ix, host.default_params(paired=1, max_mismatch=5, max_intron=max_intron))
    orc = oracle_py.Oracle(prefix)
    lib = gpu.lib
    w = None
    tot_reads = tot_rep = 0
    for seed in seeds:
        b = bench.Batch(lib, g, pairs, rlen, seed, 0.01, 0.02, spliced)
        if w is None:
            w = bench.Worker(gpu, b.n, "packed", "compact")
        w.map(b); res = w.result()
        t = time.time()
        o = orc.map_batch(orc.params(paired=1, max_mismatch=5, max_intron=max_intron), b.so.a, b.rl.a, b.seq.a, threads=cores)
        dt = time.time() - t
        try:
            common.assert_same(res, o)
            ok = True
        except AssertionError as e:
            ok = False; bad_total += 1
            print("   DIFFERENCE in batch seed %d: %s" % (seed, str(e)[:400]))
        tot_reads += b.n; tot_rep += len(o[1])
        print("%s batch seed %d: %d reads, %d reports, %d CIGAR ops, %d junction tuples: %s (oracle %.1f s on %d cores)" %
              (tag, seed, b.n, len(o[1]), len(o[2]), len(o[3]), "identical to the oracle, every field" if ok else "DIFFERENT", dt, cores), flush=True)
    print("%s: %d reads, %d reports compared" % (tag, tot_reads, tot_rep), flush=True)
    gpu.close(); orc.close()


if what in ("cfg3", "both"):
    prefix, g = bench.prepare_index("/tmp/dart_bench_cache", (gnames, glens), 0, lambda: None)
    import torch; torch.cuda.empty_cache()
    run("cfg3 (GRCh38-sized, 2x101, -mis 5)", prefix, g, 101, 0.0, 500000, [1000 + j for j in range(n_batches)], 1000000)
if what in ("cfg5", "both"):
    prefix, g = bench.prepare_index("/tmp/dart_bench_cache", (gnames, glens), 0, lambda: None, 20000)
    import torch; torch.cuda.empty_cache()
    run("cfg5 (GRCh38-sized, 2x151, 30 % spliced over 20000 planted introns, -max_intron 500000, -mis 5)", prefix, g, 151, 0.3, 500000, [1000], 1000000)
print("RESULT:", "all batches identical to the oracle" if bad_total == 0 else "%d batch(es) differ" % bad_total)
sys.exit(1 if bad_total else 0)
