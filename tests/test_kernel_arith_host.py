"""CPU suite: the pure arithmetic helpers the kernels are built from (nst_nt4 restatement, the x2 truncation of nw_alignment,
the four-characters-at-once read encoder, the RefSequence window fetches across both strand boundaries, and the serial nw_alignment restatement d_nw against the oracle's on
20 000 random pairs), compiled for the host with hipcc and run without a GPU (no HIP API call)."""
import os, shutil, subprocess
import pytest
import common, oracle_py


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_kernel_arithmetic_helpers_on_host(workdir):
    src = os.path.join(common.ROOT, "tests", "native", "host_checks.hip")
    exe = os.path.join(workdir, "host_checks")
    oracle_py.build()                                   # the checker the harness links (orc_nw)
    odir = os.path.dirname(oracle_py.LIB)
    subprocess.run(["hipcc", "-O2", "--offload-arch=gfx950", "-std=c++17", "-w", "-o", exe, src, "-L" + odir, "-loracle", "-Wl,-rpath," + odir], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert out.strip().endswith("bad=0") and "d_nw: 20000" in out, out


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_chain_stage_lane_code_on_host_against_reference_dumps(workdir):
    """dg_chain.h's per-lane code (sort, GenerateAlignmentCandidate, CheckPairedAlignmentCandidates, RemoveUnMated...,
    RemoveRedundantCandidates) compiled for the host and fed the seeds of the REFERENCE's stage dumps (S1/S2 lines): the
    candidates must be the reference's C1/C2 lines -- score, PosDiff, mate index, seed count -- for every read of every case."""
    import gzip
    src = os.path.join(common.ROOT, "tests", "native", "chain_checks.hip")
    exe = os.path.join(workdir, "chain_checks")
    subprocess.run(["hipcc", "-O2", "--offload-arch=gfx950", "-std=c++17", "-w", "-o", exe, src], check=True)
    total = 0
    for name, spec in sorted(common.MANIFEST["cases"].items()):
        run0 = common.MANIFEST["manifest"][name]["runs"][0]
        p, _ = common.parse_flags(run0["flags"])
        lengths = spec["lengths"]
        paired = 1 if spec["paired"] else 0
        inp = ["H %d %d %d %d %d" % (len(lengths), sum(lengths), p.get("max_gaps", 5), p.get("max_intron", 500000), paired)]
        o = 0
        for l in lengths:
            inp.append("%d %d" % (o, l)); o += l
        want, unit = [], {}
        for line in gzip.open(os.path.join(common.GOLDEN, run0["base"] + ".stages.gz"), "rt"):
            f = line.split()
            if f[0] in ("S1", "S2"):
                toks = [t.split(":") for t in f[3:]]
                unit[f[0]] = "%d %d " % (spec["rlen"], len(toks)) + " ".join("%s %s %s" % (t[0], t[1], t[2]) for t in toks)
                if (f[0] == "S2") or not paired:
                    inp.append("U " + unit["S1"] + (" " + unit["S2"] if paired else ""))
                    unit = {}
            elif f[0] in ("C1", "C2"):
                want.append(" ".join([f[0]] + f[2:]))
        out = subprocess.run([exe], input="\n".join(inp) + "\n", check=True, capture_output=True, text=True).stdout.strip().split("\n")
        assert len(out) == len(want) and len(want) > 100, (name, len(out), len(want))
        for got, w in zip(out, want):
            assert got == w, (name, got, w)
        total += len(want)
    assert total > 1000


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_report_list_passes_on_host_against_oracle(workdir):
    """dg_report.h's list passes (tandem / translocation clean-up, overlap trimming + normal pairs, splice junctions) compiled
    for the host and run against the oracle's restatement of the same reference functions on 660 000 random seed lists"""
    src = os.path.join(common.ROOT, "tests", "native", "report_checks.hip")
    exe = os.path.join(workdir, "report_checks")
    oracle_py.build()
    odir = os.path.dirname(oracle_py.LIB)
    subprocess.run(["hipcc", "-O2", "--offload-arch=gfx950", "-std=c++17", "-w", "-o", exe, src, "-L" + odir, "-loracle", "-Wl,-rpath," + odir], check=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip().endswith("bad=0"), out.stdout + out.stderr


def _pair_check_file(path, prefix, params, paired, reads_arr, orc):
    import numpy as np
    from dart_amd import host
    so, rl, flat = host.pack_reads(reads_arr)
    o_reads, o_rep, o_cig, o_sj = orc.map_batch(orc.params(paired=paired, **params), so, rl, flat, threads=8)
    seed_off = [0]; rp, sl, gp = [], [], []
    sp = orc.params(**{k: v for k, v in params.items() if k in ("max_dup",)})
    for i in range(len(rl)):
        a, b, c = orc.seeds(sp, reads_arr[i].tobytes() if hasattr(reads_arr[i], "tobytes") else reads_arr[i])
        rp.append(a); sl.append(b); gp.append(c); seed_off.append(seed_off[-1] + len(a))
    rp = np.concatenate(rp).astype(np.int32); sl = np.concatenate(sl).astype(np.int32); gp = np.concatenate(gp).astype(np.int64)
    ix = host.Index(prefix)
    p = dict(max_gaps=5, max_dup=100, max_intron=500000, min_intron=5, max_mismatch=0, multi_hit=0, all_sj=0); p.update(params)
    hdr = np.zeros(16, np.int32)
    hdr[:10] = [len(ix.names), paired, len(rl), p["max_gaps"], p["max_dup"], p["max_intron"], p["min_intron"], p["max_mismatch"], p["multi_hit"], p["all_sj"]]
    hdr[10] = np.int64(ix.l_pac & 0xFFFFFFFF).astype(np.uint32).view(np.int32) if False else int(np.array([ix.l_pac & 0xFFFFFFFF], np.uint32).view(np.int32)[0])
    hdr[11] = ix.l_pac >> 32
    hdr[12:16] = [len(rp), len(o_rep), len(o_cig), len(flat)]
    with open(path, "wb") as f:
        for a in (hdr, ix.chr_off.astype(np.int64), ix.chr_len.astype(np.int64), ix.pac[: ix.l_pac // 4 + 1], so.astype(np.uint32), rl.astype(np.uint16), flat,
                  np.asarray(seed_off, np.uint32), rp, sl, gp, o_reads, o_rep, o_cig):
            f.write(np.ascontiguousarray(a).tobytes())


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_pair_kernel_unit_code_on_host_against_oracle(workdir):
    """k_pair's per-unit code (dg_pair.h) compiled for the host: from the oracle's seeds of each unit to finished records,
    compared field by field with the oracle's records for every unit the code finishes itself -- the golden cases under their
    flag sets, and a repeat-rich genome where reads have several equally good candidates (pair settling, FLAG, MAPQ ties)"""
    import numpy as np
    from dart_amd import synth, index_build, host
    src = os.path.join(common.ROOT, "tests", "native", "pair_checks.hip")
    exe = os.path.join(workdir, "pair_checks")
    subprocess.run(["hipcc", "-O2", "--offload-arch=gfx950", "-std=c++17", "-w", "-o", exe, src], check=True)
    runs = []
    for name in sorted(common.MANIFEST["cases"]):
        c = common.build_case(name, workdir)
        for run in c["runs"]:
            p, _ = common.parse_flags(run["flags"])
            runs.append((c["prefix"], p, int(c["spec"]["paired"]), c["reads"]))
    g = synth.make_genome([600000, 400000], seed=71, repeat_scale=300.0, n_introns=50)
    prefix = os.path.join(workdir, "pairchk")
    index_build.build_index_from_genome(g, prefix, device="cpu")
    m1, m2 = synth.make_reads(g, 6000, rlen=101, seed=72, sub_rate=0.01, indel_frac=0.02, n_frac=0.003, spliced_frac=0.02)
    arr = host.interleave_pairs(m1, m2)
    for flags in (["-mis", "5"], [], ["-mis", "5", "-m"], ["-mis", "2", "-max_dup", "1000"]):
        p, _ = common.parse_flags(flags)
        runs.append((prefix, p, 1, arr))
    runs.append((prefix, common.parse_flags(["-mis", "4"])[0], 0, m1))
    fast = total = multi = packed = 0
    for k, (pf, p, paired, reads_arr) in enumerate(runs):
        path = os.path.join(workdir, "pairchk_%d.bin" % k)
        orc = oracle_py.Oracle(pf)
        _pair_check_file(path, pf, p, paired, reads_arr, orc)
        orc.close()
        out = subprocess.run([exe, path], capture_output=True, text=True)
        assert out.returncode == 0 and out.stdout.strip().endswith("bad=0"), (k, p, out.stdout + out.stderr)
        f = out.stdout.split()
        total += int(f[1].rstrip(":")); fast += int(f[4].rstrip(",")); multi += int(f[f.index("reports") + 1].rstrip(";"))
        packed += int(f[f.index("words") + 1].rstrip(";"))
    # (packed: units whose reads hold A/C/G/T/N only were run a second time from 2-bit + mask words -- what k_pair does for a packed batch -- and gave the same state)
    assert total > 20000 and fast > 0.5 * total and multi > 200 and packed > 0.5 * total, (total, fast, multi, packed)


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_gap_filling_without_strings_equals_the_string_form(workdir):
    """dg_report.h's two forms of FillGapsBetweenAdjacentSeeds (called by SeedExtension, AlignmentCandidates.cpp:577-594): d_gap_small -- read gaps of at most 24 bases, both
    alignments by the lane itself, the split search on bit masks -- against the gapped-string form on 400 000 random read gaps (both strands, the strand boundary, the ends of
    the text, junctions inside the gap, indels, lower case, N); every split point 0 .. 24 must have occurred, and so must read bases beyond either window."""
    src = os.path.join(common.ROOT, "tests", "native", "gap_checks.hip")
    exe = os.path.join(workdir, "gap_checks")
    subprocess.run(["hipcc", "-O2", "--offload-arch=gfx950", "-std=c++17", "-w", "-o", exe, src], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip().endswith("bad=0"), r.stdout + r.stderr
    line = [l for l in r.stdout.splitlines() if l.startswith("gap filling:")][0]
    counts = [int(x) for x in line.split("split points:")[1].split()]
    assert len(counts) == 25 and all(c > 0 for c in counts), counts
    beyond = line.split("with read bases beyond")[0].split(",")[-1].split()
    assert int(beyond[0]) > 1000 and int(beyond[2]) > 1000, line


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_traceback_consumers_on_the_bits_equal_the_string_forms(workdir):
    """dg_report.h's consumers of a wave-wide nw_alignment that walk its traceback bits (TbWalk) -- d_process_pair_tb (tools.cpp:130-164,203-300: CIGAR runs, score,
    CheckLocalAlignmentQuality, head / tail trimming) and d_gap_right_tb / d_gap_left_tb / d_gap_split_tb (FillGapsBetweenAdjacentSeeds on read gaps wider than 24 bases)
    -- against the gapped-string forms on the same bits: 60 000 segment pairs in the three modes and 30 000 wide gaps; bits from the cell recurrence, from a random
    source and from laid-out three-run paths, so that every trimming case and accepted splits occur in numbers."""
    src = os.path.join(common.ROOT, "tests", "native", "tb_checks.hip")
    exe = os.path.join(workdir, "tb_checks")
    subprocess.run(["hipcc", "-O2", "--offload-arch=gfx950", "-std=c++17", "-w", "-o", exe, src], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip().endswith("bad=0"), r.stdout + r.stderr
    import re
    pairs = [l for l in r.stdout.splitlines() if l.startswith("pairs:")][0]
    gaps = [l for l in r.stdout.splitlines() if l.startswith("wide gaps:")][0]
    n = [int(x) for x in re.findall(r"\d+", pairs)]
    assert n[0] > 50000 and min(n[1:4]) > 15000 and n[5] > 10000 and n[6] > 2000 and n[7] > 2000, pairs      # compared, per mode, quality failures, head / tail trims
    g = [int(x) for x in re.findall(r"\d+", gaps)]
    assert g[0] == 30000 and g[1] > 3000 and g[2] > 2000 and g[3] > 2000, gaps
