"""Roofline calibration: how many random 64-byte Occ blocks per second can one MI355X deliver?
(the access pattern of k_seed/k_locate with no arithmetic).  Usage: python profiles/random_block_ceiling.py [genome_len]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dart_amd import host
glen = int(sys.argv[1]) if len(sys.argv) > 1 else 64444167
prefix = '/tmp/dart_bench_cache/g%d' % glen
ix = host.Index(prefix)
gpu = host.DartGPU(ix, host.default_params(paired=1, max_mismatch=5))
f = gpu.lib.dg_debug_random_blocks
f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
print("index: %d Occ blocks of 64 B (%.0f MB)" % ((ix.seq_len + 127) // 128, (ix.seq_len + 127) // 128 * 64 / 1e6))
for dep in (1, 0, 3, 2):
    for wpc in (8, 16, 32):
        ms = C.c_float(0); n = C.c_ulonglong(0)
        rc = f(gpu.ctx, 200, dep, wpc, C.byref(ms), C.byref(n))
        assert rc == 0
        print(("quad/block " if dep >= 2 else "lane/block ") + "%s chains, %2d waves/CU: %6.1f G blocks/s = %5.2f TB/s" % ("dependent  " if dep & 1 else "independent", wpc, n.value / ms.value / 1e6, n.value * 64 / ms.value / 1e9))
