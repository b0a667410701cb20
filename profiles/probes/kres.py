#!/usr/bin/env python3
"""Per-kernel resources of a gfx950 build: python3 profiles/probes/kres.py <file.s from hipcc -save-temps>
(VGPRs as the wave allocates them, LDS, scratch, and the waves per SIMD / workgroups per CU those allow)."""
import re, sys
txt = open(sys.argv[1]).read()
meta = txt[txt.rfind("amdhsa.kernels:"):]
ents = re.split(r"\n  - ", meta)[1:]
rows = []
for e in ents:
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, e) or [None, "0"])[1]
    name = g("name")
    short = re.sub(r"^_Z\d+", "", name)[:28]
    v, a, s, lds, prv, wg = int(g("vgpr_count")), int(g("agpr_count")), int(g("sgpr_count")), int(g("group_segment_fixed_size")), int(g("private_segment_fixed_size")), int(g("max_flat_workgroup_size"))
    tot = v + a
    alloc = (tot + 7) // 8 * 8
    wps = min(8, 512 // alloc) if alloc else 8
    rows.append((short, v, a, s, lds, prv, wg, wps))
print("%-30s %5s %5s %5s %7s %7s %5s %9s" % ("kernel", "vgpr", "agpr", "sgpr", "lds", "scratch", "wg", "waves/SIMD"))
for r in sorted(rows): print("%-30s %5d %5d %5d %7d %7d %5d %9d" % r)
