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
