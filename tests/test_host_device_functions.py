"""CPU suite: __host__ __device__ functions of the kernels' headers run on the host (hipcc compiles the same source for both sides).
splice_windows_check.hip: d_identify_sj -- the splice-motif search of CheckSpliceJunction (AlignmentCandidates.cpp:732-756) with its two genome windows
fetched once into registers -- against the same search through single reference characters, on 9.6 M random junctions (both strand halves, the strand
boundary, the ends of the text, low-complexity texts where boundary shifts pass CheckSeqFragment), and d_ref_codes against d_refchar base by base."""
import os, subprocess
import common


def test_splice_motif_search_in_registers_equals_the_character_form(workdir):
    import __graft_entry__ as ge
    src = os.path.join(common.ROOT, "tests", "host", "splice_windows_check.hip")
    exe = os.path.join(workdir, "splice_windows_check")
    subprocess.check_call([ge.HIPCC, "-O2", "--offload-arch=gfx950", "-std=c++17", "-o", exe, src], stderr=subprocess.DEVNULL)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "mismatches 0" in r.stdout and "bad 0" in r.stdout, r.stdout
    # every outcome occurred: the 19 shifts and "no motif"
    counts = [int(x) for x in r.stdout.strip().splitlines()[-1].split(":")[1].split()]
    assert len(counts) == 20 and all(c > 0 for c in counts), counts
