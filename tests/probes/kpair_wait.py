"""Usage: python tests/probes/kpair_wait.py [grch38|chr20] [planted|human] [inflight]
Where does k_pair's time go when twelve batches share the GPU (4.4 ms in the timed region against 0.83 ms alone)?  Every
workgroup of the kernel leaves three time stamps in the scan trace (dg_scan.h): ticket (= start), publication of its own totals (= its units are
processed, the look-back starts), publication of its prefix (= the look-back is over; the records are written after that).  From the traces of the
contexts' last batches: per tile the time spent processing and the time spent waiting for predecessors, and how both compare with the kernel's
HIP-event duration -- alone and with N batches in flight."""
import ctypes as C, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench
from dart_amd import synth, host

label, gnames, glens = bench.genome_spec(sys.argv[1] if len(sys.argv) > 1 else "grch38")
model = sys.argv[2] if len(sys.argv) > 2 else "planted"
inflight = int(sys.argv[3]) if len(sys.argv) > 3 else 12
prefix, g = bench.prepare_index("/tmp/dart_bench_cache", (gnames, glens), 0, lambda: None, 0, 1.0, model)
import torch
torch.cuda.empty_cache()
ix = host.Index(prefix)
pairs = 1000000
lib = host._load_lib()
batches = [bench.Batch(lib, g, pairs, 101, 1000 + j, 0.01, 0.02, 0.0) for j in range(4)]       # page-locked, packed: the bench's own batches
gpu = host.DartGPU(ix, host.default_params(paired=1, max_mismatch=5))
gpu.lib.dg_debug_scan_trace.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
workers = [bench.Worker(gpu if k == 0 else gpu.clone(), batches[0].n, "packed", "compact") for k in range(inflight)]
ctxs = [w.gpu for w in workers]


def trace(ctx, which=1):
    out = np.zeros((8192, 4), np.uint64); n = C.c_size_t(0); tpm = C.c_double(0)
    rc = ctx.lib.dg_debug_scan_trace(ctx.ctx, which, out.ctypes.data, 8192, C.byref(n), C.byref(tpm))
    assert rc == 0, rc
    t = out[:int(n.value)]
    mask = np.uint64((1 << 56) - 1)
    tk, p1, p2 = t[:, 1].astype(np.float64), t[:, 2].astype(np.float64), (t[:, 3] & mask).astype(np.float64)
    ok = (p1 > 0) & (p2 > 0)
    ok[0] = False                                   # (tile 0 publishes no totals of its own)
    return tk[ok] / tpm.value, p1[ok] / tpm.value, p2[ok] / tpm.value, (t[:, 3] >> np.uint64(56))[ok], t[:, 0][ok] & np.uint64(0xFFFFFFFF)


def report(tag, ctx_list, kern_ms):
    proc, wait, span = [], [], []
    for c in ctx_list:
        tk, p1, p2, xcc, hw = trace(c)
        proc.append(p1 - tk); wait.append(p2 - p1); span.append(p2.max() - tk.min())
    proc, wait = np.concatenate(proc), np.concatenate(wait)
    q = lambda a: "mean %.3f  median %.3f  p90 %.3f  max %.3f ms" % (a.mean(), np.median(a), np.quantile(a, 0.9), a.max())
    print("%s: k_pair HIP-event time %.2f ms per launch; first ticket -> last prefix %.2f ms; %d tiles per launch" % (tag, kern_ms, float(np.mean(span)), len(proc) // len(ctx_list)))
    print("   per tile, ticket -> own totals (processing its 256 units): " + q(proc))
    print("   per tile, own totals -> prefix (waiting for predecessors):  " + q(wait))
    print("   share of the tiles' resident time spent waiting: %.1f %%" % (100 * wait.sum() / (wait.sum() + proc.sum())), flush=True)


def run_round(ws, rounds):
    kern = [0.0] * len(ws)
    def work(k):
        for i in range(rounds):
            ws[k].map(batches[(k + i) % len(batches)])
        kern[k] = dict(ws[k].gpu.timings()).get("k_pair", 0.0)
    th = [threading.Thread(target=work, args=(k,)) for k in range(len(ws))]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    print("   (%d contexts x %d batches of 2 M reads in %.1f ms = %.0f M reads/s host to host)" % (len(ws), rounds, dt * 1e3, len(ws) * rounds * 2 * pairs / dt / 1e6), flush=True)
    return float(np.mean(kern))

print(label + ", " + model + ")", flush=True)
run_round(workers, 2)                               # sizes every context's buffers
report("one batch alone", [gpu], run_round(workers[:1], 3))
report("%d batches in flight" % inflight, ctxs, run_round(workers, 8))
