"""One batch of the bench workload on a chosen genome: per-kernel times (one batch in flight), how many units took which path,
host<->device copy rates with page-locked buffers.  python profiles/probes/stage_stats.py [chr20|grch38] [pairs]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from dart_amd import host, synth
which = sys.argv[1] if len(sys.argv) > 1 else "chr20"
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
label, names, lens = bench.genome_spec(which)
prefix, g = bench.prepare_index("/tmp/dart_bench_cache", (names, lens), 0, lambda: None)
ix = host.Index(prefix)
gpu = host.DartGPU(ix, host.default_params(paired=1, max_mismatch=5))
m1, m2 = synth.make_reads(g, pairs, rlen=101, seed=1000, sub_rate=0.01, indel_frac=0.02, n_frac=0.002)
arr = host.interleave_pairs(m1, m2)
so, rl, flat = host.pack_reads(arr)
gpu.upload(so, rl, flat)
for it in range(4):
    t = time.perf_counter(); gpu.run(); dt = time.perf_counter() - t
    c = gpu.counters()
    print("run %d: %.2f ms wall; general-path units %d (%.2f %%), wave-chained units %d (%.2f %%), batch runs %d, seeds %d, cands %d" %
          (it, dt * 1e3, c["general_path_units"], 100.0 * c["general_path_units"] / pairs, c["wave_chained_units"], 100.0 * c["wave_chained_units"] / pairs, c["batch_runs"], c["seeds"], c["candidates"]))
print("kernels (ms):", " ".join("%s=%.3f" % kv for kv in gpu.timings()))
# host link: the batch in and the records out through page-locked buffers
words, nlist = host.pack_reads_2bit(arr)
n = len(rl)
import ctypes as C
p_seq = gpu.pinned((len(flat) + 64,), np.uint8); p_so = gpu.pinned((n,), np.uint32); p_rl = gpu.pinned((n,), np.uint16)
p_seq.a[:len(flat)] = flat; p_so.a[:] = so; p_rl.a[:] = rl
p_w = gpu.pinned(words.shape, np.uint32); p_w.a[:] = words; p_n = gpu.pinned((max(len(nlist), 1),), np.uint32); p_n.a[:len(nlist)] = nlist
caps = (C.c_size_t * 3)(n * 2, n * 6, n); used = (C.c_size_t * 3)()
o_r = gpu.pinned((n,), host.READ_OUT); o_p = gpu.pinned((caps[0],), host.REPORT_OUT); o_c = gpu.pinned((caps[1],), np.uint32); o_s = gpu.pinned((caps[2],), host.SJ_OUT)
for name, fn in (("ascii  ", lambda: gpu.lib.dg_map_batch(gpu.ctx, n, p_so.a.ctypes.data, p_rl.a.ctypes.data, p_seq.a.ctypes.data, o_r.a.ctypes.data, o_p.a.ctypes.data, o_c.a.ctypes.data, o_s.a.ctypes.data, caps, used)),
                 ("packed ", lambda: gpu.lib.dg_map_batch_packed(gpu.ctx, n, 101, None, words.shape[1], p_w.a.ctypes.data, p_n.a.ctypes.data, len(nlist), o_r.a.ctypes.data, o_p.a.ctypes.data, o_c.a.ctypes.data, o_s.a.ctypes.data, caps, used))):
    for it in range(3):
        t = time.perf_counter(); rc = fn(); dt = time.perf_counter() - t
        assert rc == 0, rc
    out_b = n * 36 + used[0] * 40 + used[1] * 4 + used[2] * 24
    in_b = (len(flat) + 6 * n) if name.startswith("ascii") else (words.nbytes + 4 * len(nlist))
    print("%s host-to-host, one context: %.2f ms per %d reads; in %.1f MB, out %.1f MB" % (name, dt * 1e3, n, in_b / 1e6, out_b / 1e6))
