"""which engine copies page-locked host memory <-> HBM on this box (run with AMD_LOG_LEVEL=4 and look for 'HSA Copy' / blit lines),
and how fast under each runtime switch.  Usage: python profiles/probes/copy_engine_probe.py"""
import time, torch
n = 256 << 20
h = torch.empty(n, dtype=torch.uint8, pin_memory=True)
d = torch.empty(n, dtype=torch.uint8, device="cuda")
for name, f in (("H2D", lambda: d.copy_(h, non_blocking=True)), ("D2H", lambda: h.copy_(d, non_blocking=True))):
    f(); torch.cuda.synchronize()
    t = time.time()
    for _ in range(8): f()
    torch.cuda.synchronize()
    print(name, "%.1f GB/s" % (8 * n / (time.time() - t) / 1e9), flush=True)
