"""Host link rates with page-locked buffers: H2D alone, D2H alone, both at once (is the link full duplex for us?), and D2H
split over several streams.  python profiles/probes/pcie_probe.py"""
import time, torch
assert torch.cuda.is_available()
MB = 1 << 20
def pinned(n): return torch.empty(n, dtype=torch.uint8).pin_memory()
h_in, h_out = pinned(224 * MB), pinned(676 * MB)
d_in, d_out = torch.empty(224 * MB, dtype=torch.uint8, device="cuda"), torch.empty(676 * MB, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
streams = [torch.cuda.Stream() for _ in range(8)]
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps
def h2d():
    with torch.cuda.stream(s1): d_in.copy_(h_in, non_blocking=True)
def d2h():
    with torch.cuda.stream(s2): h_out.copy_(d_out, non_blocking=True)
def both(): h2d(); d2h()
def d2h_split(k):
    def f():
        n = h_out.numel() // k
        for i in range(k):
            with torch.cuda.stream(streams[i]): h_out[i * n:(i + 1) * n].copy_(d_out[i * n:(i + 1) * n], non_blocking=True)
    return f
t = timeit(h2d); print("H2D alone   %6.1f GB/s (%.0f MB in %.2f ms)" % (h_in.numel() / t / 1e9, h_in.numel() / MB, t * 1e3))
t = timeit(d2h); print("D2H alone   %6.1f GB/s (%.0f MB in %.2f ms)" % (h_out.numel() / t / 1e9, h_out.numel() / MB, t * 1e3))
t = timeit(both); print("both at once: H2D %.0f MB + D2H %.0f MB in %.2f ms = %.1f GB/s in total" % (h_in.numel() / MB, h_out.numel() / MB, t * 1e3, (h_in.numel() + h_out.numel()) / t / 1e9))
for k in (2, 4, 8):
    t = timeit(d2h_split(k)); print("D2H over %d streams %6.1f GB/s" % (k, h_out.numel() / t / 1e9))
