"""Usage: python tests/probes/init_probe.py [grch38|chr20]
Start-up cost and mapping rate per choice of look-up aids (full SA density x prefix-table K) on the bench genome: dg_init_files' split
(allocation, files -> HBM, build kernels), then one context mapping a 1 M-pair batch a few times.  Also: does a pause between two
processes' worth of 118 GB allocations change hipMalloc's time (the driver clears VRAM that another allocation used before)?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench
from dart_amd import synth, host

label, gnames, glens = bench.genome_spec(sys.argv[1] if len(sys.argv) > 1 else "grch38")
prefix, g = bench.prepare_index("/tmp/dart_bench_cache", (gnames, glens), 0, lambda: None)
import torch
torch.cuda.empty_cache()
ix = host.Index(prefix)
pairs = 1000000
m1, m2 = synth.make_reads(g, pairs, rlen=101, seed=1000, sub_rate=0.01, indel_frac=0.02, n_frac=0.002)
arr = host.interleave_pairs(m1, m2)
words, nlist = host.pack_reads_2bit(arr)
params = host.default_params(paired=1, max_mismatch=5)
print(label + ")", flush=True)


def run(dense, K, async_aids=False, pause=0.0, reps=4):
    for k, v in (("DG_SA_DENSE", dense), ("DG_KTAB_K", K)):
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = str(v)
    if pause: time.sleep(pause)
    t0 = time.perf_counter()
    gpu = host.DartGPU(ix, params, async_aids=async_aids)
    t_init = time.perf_counter() - t0
    times = []
    kern = None
    for i in range(reps + 1 if not async_aids else 40):
        t = time.perf_counter()
        gpu.map_batch_compact(words, nlist, 101)
        dt = time.perf_counter() - t
        times.append(dt)
        kern = dict(gpu.timings())
        if async_aids and i % 4 == 3:
            print("      async: batch %d took %.1f ms (k_seed %.2f, k_locate %.2f); t = %.2f s since init began" % (i, dt * 1e3, kern.get("k_seed", 0), kern.get("k_locate", 0), time.perf_counter() - t0), flush=True)
    if async_aids: gpu.wait_index()
    best = min(times[1:])
    print("dense=%s K=%s async=%d pause=%.0f: dg_init_files %.2f s; one context, 2 M reads per batch: best %.1f ms = %.0f M reads/s (first %.0f ms); k_seed %.2f k_locate %.2f k_pair %.2f k_report %.2f ms\n      %s"
          % (dense, K, async_aids, pause, t_init, best * 1e3, 2 * pairs / best / 1e6, times[0] * 1e3, kern.get("k_seed", 0), kern.get("k_locate", 0), kern.get("k_pair", 0), kern.get("k_report", 0), gpu.init_report()), flush=True)
    gpu.close()


run(None, None)                 # full aids (full SA, K = 16 on the GRCh38-sized text)
run(None, None)                 # again at once: the 118 GB this process just freed
run(None, None, pause=8.0)      # and after a pause
for dense, K in ((1, 14), (4, 14), (8, 14), (0, 14), (4, 13), (4, 12), (0, 0), (1, 0), (0, 16)):
    run(dense, K)
run(None, None, async_aids=True)
run(4, 14, async_aids=True)
