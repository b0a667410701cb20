"""Histogram of rocprofv3 PC samples (profiles/probes/pc_sampling.sh): per kernel, the source lines (line tables) and instructions the sampled waves were at."""
import sys, os, csv, collections, re
root = sys.argv[1]
pcs = [os.path.join(d, f) for d, _, fs in os.walk(root) for f in fs if "pc_sampling" in f and f.endswith(".csv")]
kts = [os.path.join(d, f) for d, _, fs in os.walk(root) for f in fs if "kernel_trace" in f and f.endswith(".csv")]
disp = {}
for kt in kts:
    with open(kt, newline="") as fh:
        for row in csv.DictReader(fh):
            k = row.get("Dispatch_Id") or row.get("Dispatch_ID")
            if k: disp[k] = re.sub(r"\(.*", "", row.get("Kernel_Name", "?"))[:40]
csv.field_size_limit(1 << 30)
per_kernel = collections.Counter(); by_line = collections.defaultdict(collections.Counter); by_inst = collections.defaultdict(collections.Counter); extra = collections.defaultdict(collections.Counter)
n = 0
for pc in pcs:
    with open(pc, newline="") as fh:
        rd = csv.DictReader(fh)
        for row in rd:
            n += 1
            kern = disp.get(row.get("Dispatch_Id", ""), "dispatch " + row.get("Dispatch_Id", "?"))
            per_kernel[kern] += 1
            com = row.get("Instruction_Comment", "")
            m = re.search(r"(dg_\w+\.(?:h|hip):\d+)", com)
            by_line[kern][m.group(1) if m else (com[-60:] or "?")] += 1
            by_inst[kern][row.get("Instruction", "?").split(" ")[0]] += 1
            for col in ("Stall_Reason", "Wave_Issued", "Instruction_Type", "Snapshot_Stall_Reason"):
                if col in row and row[col] != "": extra[kern][col + "=" + row[col]] += 1
print("samples:", n, "files:", pcs)
for kern, c in per_kernel.most_common(12):
    print("\n== %s: %d samples (%.1f %%)" % (kern, c, 100.0 * c / max(n, 1)))
    print("  opcodes: " + "  ".join("%s %.1f%%" % (k, 100.0 * v / c) for k, v in by_inst[kern].most_common(12)))
    if extra[kern]: print("  " + "  ".join("%s %.1f%%" % (k, 100.0 * v / c) for k, v in extra[kern].most_common(16)))
    for ln, v in by_line[kern].most_common(45): print("  %6.2f %%  %s" % (100.0 * v / c, ln))
