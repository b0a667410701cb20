"""Usage: python profiles/concurrency.py <rocprofv3 output dir> [last_seconds]
What runs beside what when a dozen batches share the GPU: from the kernel trace (`rocprofv3 --kernel-trace --output-format csv`: one row per
dispatch with start / end time stamps) of the last `last_seconds` of the run (default 1.0: the timed region of a short bench run):
  * the fraction of the time no kernel, 1, 2, ... kernels are running (concurrency histogram),
  * per kernel: dispatches, busy time (sum of durations), mean duration, its share of the kernel-seconds, VGPRs / LDS / workgroup size as the
    trace reports them (what bounds how many of its waves fit a CU),
  * kernel-seconds per wall-second (= mean concurrency)."""
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
last = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = []
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
if not rows:
    sys.exit("no *kernel_trace.csv under " + root)
ev = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    ev.append((s, e, r["Kernel_Name"].split("(")[0].replace("void ", "")[:48], r))
t_end = max(e for _, e, _, _ in ev)
t0 = t_end - int(last * 1e9)
ev = [x for x in ev if x[1] > t0]
pts = []
for s, e, n, _ in ev:
    pts.append((max(s, t0), 1)); pts.append((e, -1))
pts.sort()
hist = defaultdict(int)
cur, prev = 0, t0
for t, d in pts:
    hist[cur] += t - prev
    prev = t; cur += d
wall = t_end - t0
print("window: last %.3f s of the trace, %d dispatches" % (wall / 1e9, len(ev)))
print("concurrency histogram (share of the wall time with k kernels running):")
for k in sorted(hist):
    if hist[k] / wall >= 0.002:
        print("  %2d kernels: %5.1f %%" % (k, 100.0 * hist[k] / wall))
busy = defaultdict(lambda: [0, 0, None])
for s, e, n, r in ev:
    b = busy[n]; b[0] += 1; b[1] += e - max(s, t0); b[2] = r
tot = sum(b[1] for b in busy.values())
print("kernel-seconds per wall-second (mean concurrency): %.2f" % (tot / wall))
print("%-48s %6s %10s %9s %6s  %s" % ("kernel", "calls", "busy ms", "mean ms", "share", "VGPR / AGPR / SGPR / LDS B / workgroup / grid"))
for n, (c, t, r) in sorted(busy.items(), key=lambda kv: -kv[1][1]):
    if t / tot < 0.002:
        continue
    g = lambda k: r.get(k, "?")
    print("%-48s %6d %10.2f %9.3f %5.1f%%  %s / %s / %s / %s / %s / %s" % (n, c, t / 1e6, t / c / 1e6, 100.0 * t / tot, g("VGPR_Count"), g("Accum_VGPR_Count"), g("SGPR_Count"), g("LDS_Block_Size"),
                                                                       g("Workgroup_Size_X") if "Workgroup_Size_X" in r else g("Workgroup_Size"), g("Grid_Size_X") if "Grid_Size_X" in r else g("Grid_Size")))

# ---- per hardware queue: how busy is it, and where are the gaps?  (a context's stream runs on one hardware queue; several streams can share one:
#      a kernel then waits for another context's kernel ahead of it in the same queue)
byq = defaultdict(list)
for s_, e_, n_, r_ in ev:
    byq[r_.get("Queue_Id", "?")].append((s_, e_, n_))
print("hardware queues seen: %d" % len(byq))
gap_by = defaultdict(lambda: [0, 0])
tot_gap = tot_busy = 0
for q, lst in sorted(byq.items()):
    lst.sort()
    busy_q = sum(e_ - max(s_, t0) for s_, e_, _ in lst)
    gaps = 0
    overl = 0
    for (s0, e0, n0), (s1, e1, n1) in zip(lst, lst[1:]):
        g_ = s1 - e0
        if g_ > 0:
            gaps += g_; gap_by[(n0, n1)][0] += 1; gap_by[(n0, n1)][1] += g_
        else:
            overl += 1
    tot_gap += gaps; tot_busy += busy_q
    print("  queue %-6s dispatches %5d  busy %5.1f %%  idle between its dispatches %5.1f %%  (overlapping successors: %d)" % (q, len(lst), 100.0 * busy_q / wall, 100.0 * gaps / wall, overl))
print("largest idle time between consecutive dispatches of one queue, by (previous kernel -> next kernel):")
for (a, b), (c, g_) in sorted(gap_by.items(), key=lambda kv: -kv[1][1])[:16]:
    print("  %-28s -> %-28s  %5d times, mean %7.3f ms, total %8.2f ms" % (a[:28], b[:28], c, g_ / c / 1e6, g_ / 1e6))
